# Round 6: the barrier WAR fix under the delayed-NMS stress, and what it costs. Libraries:
#   libctpn_hip.so       the fix: every persistent conv kernel in the AHEAD form (explicit read-ahead, lgkmcnt(0) in front of every barrier)
#   libctpn_hip_exp4.so  the fix the plain way (make OBJDIR=build_exp4 TARGET=../libctpn_hip_exp4.so EXTRA=-DC3_EXP_PLAIN): the old kernel forms + lgkmcnt(0)
#   libctpn_hip_old.so   the library before the fix
set -x
mkdir -p gpurun_out/r6n
export CTPN_NO_TORCH=1
V='bf16:nms_prefix=0,debug_hog=5000 bf16:nms_prefix=0,debug_hog=7000 split:tail_confine=0,nms_prefix=0 split:tail_confine=0,nms_prefix=0,debug_hog=3000 fp32:nms_prefix=0,debug_hog=20000'
for lib in ${LIBS:-libctpn_hip.so libctpn_hip_exp4.so libctpn_hip_old.so}; do
  echo "== $lib"
  CTPN_LIB_PATH=$PWD/text-detection-ctpn_amd/$lib timeout 900 python tools/r6_pipeline_race.py --reps 24 --batch 32 --variants $V 2>&1 | cut -c1-200
done > gpurun_out/r6n/war_fix_race.txt 2>&1
cat gpurun_out/r6n/war_fix_race.txt
for r in 1 2 3; do
for lib in ${BLIBS:-libctpn_hip_old.so libctpn_hip.so libctpn_hip_exp4.so}; do
  echo "== $lib"
  timeout 300 python tools/quick_bench.py --variant "lib=text-detection-ctpn_amd/$lib" --variant "lib=text-detection-ctpn_amd/$lib precision=split" --steps 40 --rounds 1 2>&1 | grep -E "^round"
done; done > gpurun_out/r6n/war_fix_bench.txt 2>&1
cat gpurun_out/r6n/war_fix_bench.txt
