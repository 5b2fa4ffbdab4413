# soak of the fused forms: many repeats of one step, every conformer compared with the row path (tools/debug_race.py)
for cfg in ${RACE_CFGS:-1024:4:6000 1024:6:6000 512:6:6000 2048:6:2000 1024:5:3000 1024:7:3000 1024:1:2000 640:6:4000}; do
set -- ${cfg//:/ }
echo "== B $1 mode $2 reps $3"
timeout 1500 python tools/debug_race.py $1 $2 $3 2>&1 | grep "rep" | grep -v "conformers beyond 2e-6: \[\] (0)" | grep -v FORWARD | awk '{n++; if (n<=6) print} END {print "   failures:", n+0}'
done
echo "== done"
