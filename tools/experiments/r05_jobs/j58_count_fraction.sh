# round 5, job 58: count / n of the 16-bit / float tile stage in three instructions (bit for bit the division): tests, A/B
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_fused.py tests/test_gpu_api.py tests/test_gpu_parity.py -x -q -m gpu -k "uint16 or float or 16 or wide or frames" 2>&1 | tail -3
{
echo "== C3 uint16"; bash tools/ab.sh --no-secondary --bits 16
echo "== C3 float32"; bash tools/ab.sh --no-secondary --bits 32
} > gpurun_out/r5j58_count_fraction.txt 2>&1
cat gpurun_out/r5j58_count_fraction.txt
