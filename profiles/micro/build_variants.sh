# builds variants of libepnet_hip.so with different -D flags into scratch/libs/
set -eu
cd "$(dirname "$0")/../../epnet_amd/csrc"
mkdir -p ../../scratch/libs
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -mllvm -amdgpu-atomic-optimizer-strategy=None -I../../include -I."
i=0
while [ $# -gt 0 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc $F $flags -c ${SRC:-fps}.hip -o /tmp/${SRC:-fps}_$name.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../scratch/libs/lib_$name.so /tmp/${SRC:-fps}_$name.o $(ls ../lib/obj/*.o | grep -v "/${SRC:-fps}.o" | tr "\n" " ")
done
