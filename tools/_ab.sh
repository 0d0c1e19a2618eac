mkdir -p gpurun_out
B="python bench.py --arch resnet50 --moped --batch 128 --steps 16 --warmup 16 --lanes 16 --repeats 3 --no-extras --no-traffic --no-cpu-baseline --no-launch-timing"
H="python bench.py --steps 20 --warmup 20 --repeats 3 --no-extras --no-traffic --no-cpu-baseline --no-launch-timing"
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step_runs"])'
(
timeout 700 python -m pytest tests -m gpu -x -q -k "fused_noise or lanes_equal or at_size or fuse or model or backward" 2>&1 | tail -3
echo "===== g8t_j direct"
BTX_LIB=build_variants/libbtx_g8t_j.so timeout 120 python tools/gpu_diag.py g8trace --prec bf16 --shape 256,1024,14,1,1 --bs 512 --warm 40 2>&1 | grep "clock\|waves\|gap"
BTX_LIB=build_variants/libbtx_g8t_j.so timeout 120 python tools/gpu_diag.py g8trace --prec bf16 --shape 512,128,28,1,1 --bs 512 --warm 40 --bare 2>&1 | grep "clock\|waves\|gap"
echo "===== g8t_j staged"
BTX_G8_STAGED=1 BTX_LIB=build_variants/libbtx_g8t_j.so timeout 120 python tools/gpu_diag.py g8trace --prec bf16 --shape 256,1024,14,1,1 --bs 512 --warm 40 2>&1 | grep "clock\|waves\|gap"
echo "== kbench direct vs staged (tune lib)"
BTX_LIB=build_variants/libbtx_tune.so timeout 300 python tools/kbench.py --bs 512 --shapes 256,1024,14,1,1 512,128,28,1,1 1024,256,14,1,1 128,512,28,1,1 512,2048,7,1,1 256,512,56,2,1 --env - BTX_G8_STAGED=1 2>&1 | grep -v Warn | tail -14
echo "== cfg5 g8base"; BTX_LIB=build_variants/libbtx_g8base.so timeout 200 $B 2>&1 | tail -1 | python -c "$P"
echo "== cfg5 default"; timeout 200 $B 2>&1 | tail -1 | python -c "$P"
echo "== cfg5 dmarpre"; BTX_LIB=build_variants/libbtx_dmarpre.so timeout 200 $B 2>&1 | tail -1 | python -c "$P"
echo "== cfg5 default"; timeout 200 $B 2>&1 | tail -1 | python -c "$P"
echo "== head default"; timeout 200 $H 2>&1 | tail -1 | python -c "$P"
echo "== head dmarpre"; BTX_LIB=build_variants/libbtx_dmarpre.so timeout 200 $H 2>&1 | tail -1 | python -c "$P"
echo "== head default"; timeout 200 $H 2>&1 | tail -1 | python -c "$P"
echo "== head dmarpre"; BTX_LIB=build_variants/libbtx_dmarpre.so timeout 200 $H 2>&1 | tail -1 | python -c "$P"
) > gpurun_out/ab.log 2>&1
cat gpurun_out/ab.log
