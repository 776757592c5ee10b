for LIB in "" gcsa2_amd/lib/libgcsa2_hip_lb5.so ""  gcsa2_amd/lib/libgcsa2_hip_lb5.so; do
  GCSA2_HIP_LIB=$LIB python bench.py --workload linear --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('lib=[$LIB]', 'linear', '%.4g q/s'%d['value'], 'kernel_ms %.4f'%d['roofline']['kernel_ms'])"
done
for LIB in "" gcsa2_amd/lib/libgcsa2_hip_lb5.so; do
  GCSA2_HIP_LIB=$LIB python bench.py --steps 10 --warmup 2 --no-cpu --no-secondary 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('lib=[$LIB]', 'human', '%.4g q/s'%d['value'], 'kernel_ms %.4f'%d['roofline']['kernel_ms'])"
done
