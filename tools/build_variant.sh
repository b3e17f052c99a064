#!/bin/bash
# Build an experimental variant of libvaeq_hip.so in which some translation units are replaced:
#   tools/build_variant.sh <name> <replacement.hip>...   (each replaces the csrc file of the same base name; extra hipcc flags via HIPFLAGS)
# -> gpurun_variants/libvaeq_<name>.so ; run with VAEQ_LIB=$PWD/gpurun_variants/libvaeq_<name>.so (tools/ab.sh A/Bs such builds).
# The unchanged units are taken from vae_equalizer_amd/_obj (python -c "from vae_equalizer_amd import _native; _native.build()").
set -e
ROOT=$(cd $(dirname $0)/.. && pwd); C=$ROOT/vae_equalizer_amd/csrc; O=$ROOT/vae_equalizer_amd/_obj; V=$ROOT/gpurun_variants
name=$1; shift
mkdir -p $V/$name.obj
objs=$(ls $O/*.o)
for src in "$@"; do
  b=$(basename $src .hip)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $HIPFLAGS -I $ROOT/include -I $C -c $src -o $V/$name.obj/$b.o
  objs=$(echo "$objs" | grep -v "/$b.o$"; echo $V/$name.obj/$b.o)
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared $objs -lhipfft -o $V/libvaeq_$name.so
rm -rf $V/$name.obj
ls -la $V/libvaeq_$name.so | awk '{print $5, $9}'
