# (the packed tooling build: NNHIP_LIB_NAME=libnewtonnet_hip_pk.so bash newtonnet_amd/csrc/build.sh -DM2_PACKED_FP32)
# final soak of the committed tree + speed A/B of molfuse2 with and without packed fp32
bash tools/race_ab.sh
echo "== speed, committed (no packed fp32 in molfuse2.hip)"
AB_MODES="0,6,4,7" timeout 900 python tools/bench_mol_fused.py 2>&1 | grep "B="
echo "== speed, molfuse2.hip WITH packed fp32 (tooling build)"
NNHIP_ALLOW_TOOLING_LIB=1 NNHIP_LIB_NAME=libnewtonnet_hip_pk.so AB_MODES="0,6,4,7" timeout 900 python tools/bench_mol_fused.py 2>&1 | grep "B="
echo "== packed build, soak (expect failures)"
RACE_LIB=pk RACE_CFGS="1024:4:4000 1024:7:8000" bash tools/race_fz.sh
