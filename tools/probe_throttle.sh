# What the driver reports (power, clocks, temperatures) while a command keeps the GPU busy:
#   bash tools/probe_throttle.sh <label> <command...>        (on the GPU box; output under gpurun_out/throttle/)
label=$1; shift
mkdir -p gpurun_out/throttle
( "$@" > gpurun_out/throttle/$label.out 2>&1 ) &
pid=$!
sleep 5
amd-smi metric -g 0 -p -c -t > gpurun_out/throttle/$label.smi 2>&1
wait $pid
echo "== $label"; tail -n 1 gpurun_out/throttle/$label.out
grep -iE "socket_power|gfx_0|CLK|clk:|mem_0|hotspot|edge|mem:|TEMPER|deep|locked" -A0 gpurun_out/throttle/$label.smi | head -60
