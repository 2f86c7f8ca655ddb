#!/bin/bash
# tools/build_variant.sh NAME [extra hipcc flags...]
# A/B builds of the kernels: compiles navtex_amd/csrc/*.hip with extra flags and links
# tools/_bin/libnavtex_amd_NAME.so from the product's host objects (build the product first).
# Select it with NAVTEX_AMD_LIB=tools/_bin/libnavtex_amd_NAME.so.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p $R/tools/_bin
O=$R/navtex_amd/_obj
kobjs=""
for src in $R/navtex_amd/csrc/*.hip; do
    o=$R/tools/_bin/$(basename $src .hip)_$name.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fvisibility=hidden -I$R/include -I$R/navtex_amd/csrc "$@" -c $src -o $o
    kobjs="$kobjs $o"
done
objs=$(ls $O/*.o | grep -v "\.hip\.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs $kobjs -o $R/tools/_bin/libnavtex_amd_$name.so -lpthread -ldl
echo $R/tools/_bin/libnavtex_amd_$name.so
