for w in 16 8; do
  echo "== brute waves $w"
  EPNET_FPS_PRUNE=0 EPNET_FPS_WAVES=$w timeout -k 10 120 python scratch/bench_fps.py 2>&1 | grep "plain.*B= 16 N=  4096\|plain.*B=256 N=  4096"
done
