# needs a library built with MMDX_BUILD_ABLATE=1
for t in ${THREADS_TO_TRY:-512 256}; do for g in ${GROUPS_TO_TRY:-8 12 16}; do for a in ${ABLATES:-0 1 4}; do
  echo -n "threads=$t group=$g ablate=$a: "
  MMDX_THREADS=$t MMDX_ABLATE=$a MMDX_GROUP=$g python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('kernel_ms %.4f  ms/step %.4f'%(r['avg_kernel_ms'], d['ms_per_step']))"
done; done; done
