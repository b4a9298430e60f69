for p in "swimmer midpoint 250 filter 1000 1e4 2000" "truck_trailer trapeze 250 filter 1000 1e6 1000" "moonlander midpoint 250 filter 1000 1e2 1000" "moonlander midpoint 250 filter 1000 1e6 1000" "space_shuttle trapeze 250 filter 1000 1e2 1000" "space_shuttle trapeze 250 filter 1000 1e6 1000" "swimmer midpoint 250 merit 1000 1e4 2000"; do
  echo "$p: $(timeout -k 10 120 python -u bench/explore_ipm_one.py $p 2>&1 | grep "elastic:" | cut -c1-330)"
done
