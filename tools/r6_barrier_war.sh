# Round 6: the delayed-NMS stress that exposed the conv kernels' barrier race (profiles/r06_barrier_war.txt), and a same-box throughput A/B of
# libraries. Default: the product library only. The A/B on file compared (LIBS / BLIBS = names under text-detection-ctpn_amd/):
#   libctpn_hip.so       the fix: every persistent conv kernel in the AHEAD form (explicit read-ahead, lgkmcnt(0) in front of every barrier)
#   libctpn_hip_exp4.so  the fix the plain way: the old kernel forms + lgkmcnt(0) (a build with AH = !FLAT && TW == 32 && !HT32 in c3_launch_p)
#   libctpn_hip_old.so   the library of the commit before the fix
set -x
mkdir -p gpurun_out/r6n
export CTPN_NO_TORCH=1
V='bf16:nms_prefix=0,debug_hog=5000 bf16:nms_prefix=0,debug_hog=7000 split:tail_confine=0,nms_prefix=0 split:tail_confine=0,nms_prefix=0,debug_hog=3000 fp32:nms_prefix=0,debug_hog=20000'
for lib in ${LIBS:-libctpn_hip.so}; do
  echo "== $lib"
  CTPN_LIB_PATH=$PWD/text-detection-ctpn_amd/$lib timeout 900 python tools/r6_pipeline_race.py --reps 24 --batch 32 --variants $V 2>&1 | cut -c1-200
done > gpurun_out/r6n/war_fix_race.txt 2>&1
cat gpurun_out/r6n/war_fix_race.txt
for r in ${ROUNDS:-1 2 3}; do
for lib in ${BLIBS:-libctpn_hip.so}; do
  echo "== $lib"
  timeout 300 python tools/quick_bench.py --variant "lib=text-detection-ctpn_amd/$lib" --variant "lib=text-detection-ctpn_amd/$lib precision=split" --steps 40 --rounds 1 2>&1 | grep -E "^round"
done; done > gpurun_out/r6n/war_fix_bench.txt 2>&1
cat gpurun_out/r6n/war_fix_bench.txt
