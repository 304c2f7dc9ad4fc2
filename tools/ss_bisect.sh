for s in 1 2 3 4 5 6 0; do
  MGS_SS_STOP=$s python tools/gpu_probe.py 2>&1 | grep "sort-only" | sed "s/^/stop=$s /"
done
