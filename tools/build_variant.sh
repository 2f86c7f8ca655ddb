#!/bin/bash
# tools/build_variant.sh NAME [extra hipcc flags...]
# A/B builds of the kernels: compiles navtex_amd/csrc/nvx_kernels.hip with extra flags and links
# tools/_bin/libnavtex_amd_NAME.so from the product's other objects (build the product first).
# Select it with NAVTEX_AMD_LIB=tools/_bin/libnavtex_amd_NAME.so.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p $R/tools/_bin
O=$R/navtex_amd/_obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fvisibility=hidden -I$R/include -I$R/navtex_amd/csrc "$@" \
    -c $R/navtex_amd/csrc/nvx_kernels.hip -o $R/tools/_bin/nvx_kernels_$name.o
objs=$(ls $O/*.o | grep -v nvx_kernels.hip.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs $R/tools/_bin/nvx_kernels_$name.o \
    -o $R/tools/_bin/libnavtex_amd_$name.so -lpthread -ldl
echo $R/tools/_bin/libnavtex_amd_$name.so
