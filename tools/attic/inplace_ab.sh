for w in ${WL:-c3_clustered c3_trainedlike c3}; do
  for W in ${WS:-0 4}; do
    GSX_REF_IN_PLACE=$W python bench.py --workload $w --test-lib --no-cpu-baseline --camera-path none > gpurun_out/ip_${w}_$W.json 2>gpurun_out/ip_${w}_$W.err
  done
done
