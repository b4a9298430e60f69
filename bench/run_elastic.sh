for p in "truck_trailer trapeze 250 filter 1000 1e2 1000" "truck_trailer midpoint 250 filter 1000 1e4 1000" "truck_trailer gauss_legendre_2 250 filter 1000 1e4 1000" "truck_trailer trapeze 250 filter 1000 1e8 1000" "truck_trailer trapeze 100 filter 1000 1e4 1000"; do
  echo "$p: $(timeout -k 10 120 python -u bench/explore_ipm_one.py $p 2>&1 | grep "elastic:" | cut -c1-330)"
done
