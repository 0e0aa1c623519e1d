for w in 1 2 4 8; do
  echo "== waves $w"
  EPNET_FPS_WAVES=$w timeout -k 10 120 python scratch/bench_fps.py 2>&1 | grep "plain.*B= 16 N=  1024\|plain.*B= 16 N=   256\|plain.*B=256 N=  1024"
done
