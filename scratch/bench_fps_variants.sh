for v in "$@"; do
  echo "== $v"
  EPNET_HIP_LIB=scratch/libs/lib_$v.so timeout -k 10 120 python scratch/bench_fps.py 2>&1 | grep "B= 16 N= 16384\|B= 16 N=  4096"
done
