# Samples board power and clocks while a command runs: bash tools/power_probe.sh <label> <command...>
label=$1; shift
mkdir -p gpurun_out/power
( "$@" > gpurun_out/power/$label.out 2>&1 ) &
pid=$!
sleep 4
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk" | tr '\n' ' ' >> gpurun_out/power/$label.smi; echo >> gpurun_out/power/$label.smi
  sleep 1
done
wait $pid
echo "== $label"; cat gpurun_out/power/$label.smi; tail -2 gpurun_out/power/$label.out
