for v in none layernorm groupnorm temporal_attn groupnorm_stats xattn none; do
  RCDM_DROP_OPS=$v timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('drop=$v', d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done
