for v in 0 1; do
  for w in spmotif_b0.9_causalgcn_h128_l3_bs128 nci1like_causalgcn_h128_l3_bs512 mutaglike_causalgat_h128_l3_bs64 spmotif_b0.9_causalgat_h128_l3_bs128 spmotif_b0.9_causalgin_h128_l3_bs128; do
    CAL_AMD_STRIPED=$v python bench.py --workload $w --steps 200 --warmup 20 --no-e2e --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('CAL_AMD_STRIPED=$v', d['config']['workload'], round(d['ms_per_step'],4), 'ms/step', 'loss', d['config'].get('final_loss'))"
  done
done
