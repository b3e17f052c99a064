#!/bin/bash
# Snapshot the current DP wave kernel sources as a named variant library: tools/snap_variant.sh <name> [extra hipcc flags]
#  -> gpurun_variants/libvaeq_<name>.so built from a copy of csrc/vaeq_dp_wave*.{hip,h}, vaeq_wave.h, vaeq_common.h (other units from vae_equalizer_amd/_obj)
set -e
ROOT=$(cd $(dirname $0)/.. && pwd); C=$ROOT/vae_equalizer_amd/csrc; name=$1; shift
D=/tmp/var/$name; rm -rf $D; mkdir -p $D
cp $C/vaeq_dp_wave.hip $C/vaeq_dp_wave_mw.hip $C/vaeq_dp_wave_mw8.hip $C/vaeq_dp_wave_kernel.h $C/vaeq_wave.h $C/vaeq_common.h $D/
HIPFLAGS="$*" $ROOT/tools/build_variant.sh $name $D/vaeq_dp_wave.hip
