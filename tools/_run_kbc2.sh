set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_stepper.py -q -m gpu -k "d3q27" -x 2>&1 | tail -5
python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "config5 or d3q27" -x 2>&1 | tail -5
B="python bench.py --workload periodic --size 384 --lattice D3Q27 --collision KBC --omega 1.9 --steps 100 --warmup 10 --cpu-baseline-seconds 0"
for pol in FP64FP32 FP32FP32; do
  for o in "" "--opt fuse2=0" "--opt exact_math=1"; do
    echo "== $pol $o"; $B --policy $pol $o 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['kernel'])"
  done
done
