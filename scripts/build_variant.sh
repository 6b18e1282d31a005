# dev tool: build_variants/lib_NAME.so = the library with la_conv_bf16.hip compiled under extra flags.  usage: build_variant.sh NAME "-DLA_ABLATE=11"
set -e
cd "$(dirname "$0")/../latentaugment_amd/csrc"
mkdir -p ../../build_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 $2 -c la_conv_bf16.hip -o /tmp/la_conv_bf16_$1.o
OBJS=$(ls *.o | grep -v la_conv_bf16.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_variants/lib_$1.so $OBJS /tmp/la_conv_bf16_$1.o
