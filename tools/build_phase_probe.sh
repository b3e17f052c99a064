#!/bin/bash
# Build a library variant with wall-clock / shader-clock stamps around the kernel phases (tools/patches/*.patch) for the phase probes:
#   tools/build_phase_probe.sh dp_wave|nn|epilogue   ->  gpurun_variants/libvaeq_<name>prof.so
#   VAEQ_LIB=$PWD/gpurun_variants/libvaeq_<name>prof.so python tools/probe_{dp,nn,epi}_phases.py        (on the GPU box)
set -e
ROOT=$(cd $(dirname $0)/.. && pwd); C=$ROOT/vae_equalizer_amd/csrc; V=$ROOT/gpurun_variants/$1prof
rm -rf $V && mkdir -p $V && cp $C/*.hip $C/*.h $V/
case $1 in dp_wave) f=vaeq_dp_wave_kernel.h;; nn) f=vaeq_nn.hip;; epilogue) f=vaeq_epilogue.hip;; *) echo "dp_wave|nn|epilogue"; exit 1;; esac
X=""
if [ $1 = nn ]; then X="-DVAEQ_NN_STAMPS -DVAEQ_NN_HALF_STAMPS"; else patch -s $V/$f $ROOT/tools/patches/$1_phase_stamps.patch; fi   # the VAE-NN kernels carry their stamps as macros
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $X -I $ROOT/include -I $V $V/*.hip -lhipfft -o $ROOT/gpurun_variants/libvaeq_$1prof.so
ls -la $ROOT/gpurun_variants/libvaeq_$1prof.so | awk '{print $5, $9}'
