for w in 4 8 2; do
  echo "== pwaves $w"
  EPNET_FPS_PWAVES=$w timeout -k 10 120 python scratch/bench_fps.py 2>&1 | grep "plain.*B= 16 N=  4096\|plain.*B= 16 N= 16384"
done
