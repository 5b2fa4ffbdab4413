# soak of a tooling build: RACE_LIB=<name> RACE_CFGS="B:mode:reps ..."
for cfg in ${RACE_CFGS:-1024:4:6000 1024:7:6000}; do
set -- ${cfg//:/ }
echo "== lib ${RACE_LIB} B $1 mode $2 reps $3"
NNHIP_ALLOW_TOOLING_LIB=1 NNHIP_LIB_NAME=libnewtonnet_hip_${RACE_LIB}.so timeout 1500 python tools/debug_race.py $1 $2 $3 2>&1 | grep "rep" | grep -v "conformers beyond 2e-6: \[\] (0)" | grep -v FORWARD | awk '{n++; if (n<=4) print} END {print "   failures:", n+0}'
done
