#!/bin/bash
# Build an experimental variant of libvaeq_hip.so with a replacement for vaeq_dp_wave.hip:  tools/build_variant.sh <name> <wave_src>
# -> gpurun_variants/libvaeq_<name>.so ; run with VAEQ_LIB=$PWD/gpurun_variants/libvaeq_<name>.so
set -e
ROOT=$(cd $(dirname $0)/.. && pwd); C=$ROOT/vae_equalizer_amd/csrc
mkdir -p $ROOT/gpurun_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I $ROOT/include -I $C $C/vaeq_dp.hip $2 $C/vaeq_awgn.hip $C/vaeq_misc.hip -o $ROOT/gpurun_variants/libvaeq_$1.so 2>&1 | grep -E "error" || true
ls -la $ROOT/gpurun_variants/libvaeq_$1.so | awk '{print $5, $9}'
