for g in ${GROUPS_TO_TRY:-8 16 22}; do
  echo -n "group=$g: "; MMDX_GROUP=$g python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('ms/step %.4f kernel_ms %.4f GB/s %.0f frac %.3f'%(d['ms_per_step'], r['avg_kernel_ms'], r['achieved'], r['frac']))"
done
