#!/usr/bin/env python3
"""Scan gfx950 assembly (hipcc --cuda-device-only -S) for the two patterns that cost this library most and are invisible in the source:
  * reads sunk into exec-masked branches of their own (`s_cbranch_execz` next to `ds_read` / `global_load` + `s_waitcnt ...cnt(0)`): the backend does this to a
    read from a clamped address that feeds a select -- one exposed round trip per operand;
  * `s_waitcnt lgkmcnt(0)` / `vmcnt(0)` per read instead of per batch.
Per kernel: instructions, exec-branches, LDS reads, lgkmcnt(0) waits, global/buffer loads, vmcnt(0) waits; flagged when waits(0) > 40 % of the reads and > 12.
usage: tools/scan_isa.py file.s [kernel-name-substring]"""
import re, sys, subprocess

def kernels(path):
    cur, body = None, {}
    for l in open(path):
        m = re.match(r'^(_Z\S+):', l)
        if m:
            cur = m.group(1); body[cur] = []; continue
        if l.startswith('.Lfunc_end'):
            cur = None; continue
        if cur and l.startswith('\t') and not l.startswith('\t.') and not l.lstrip().startswith(';'):
            body[cur].append(l.strip())
    return body

def demangle(n):
    try:
        return subprocess.run(['c++filt', n], capture_output=True, text=True).stdout.strip() or n
    except Exception:
        return n

for path in sys.argv[1:2]:
    sub = sys.argv[2] if len(sys.argv) > 2 else ''
    for name, ins in kernels(path).items():
        if sub not in name or len(ins) < 50:
            continue
        ops = [i.split()[0] for i in ins]
        nbr = sum(o.startswith('s_cbranch_exec') for o in ops)
        nds = sum(o.startswith('ds_read') for o in ops)
        nl0 = sum(i.startswith('s_waitcnt') and 'lgkmcnt(0)' in i for i in ins)
        nvm = sum(o.startswith(('global_load', 'buffer_load', 'flat_load')) for o in ops)
        nv0 = sum(i.startswith('s_waitcnt') and 'vmcnt(0)' in i for i in ins)
        flag = ' <==' if (nl0 > 12 and nl0 > 0.4 * max(nds, 1)) or (nv0 > 8 and nv0 > 0.5 * max(nvm, 1)) else ''
        print(f"{len(ins):6d} instr  exec-br {nbr:4d}  ds_read {nds:4d}  lgkm(0) {nl0:4d}  vmem-ld {nvm:3d}  vm(0) {nv0:3d}  {demangle(name)[:110]}{flag}")
