# sample power / clocks while the train step loops (GPU box)
python bench.py --no-extras --steps 1500 --warmup 5 > gpurun_out/power_bench.json 2>/dev/null &
pid=$!
sleep 14
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks --showtemp --showuse 2>/dev/null | grep -E 'Power|sclk|mclk|fclk|Temperature \(Sensor (edge|junction|memory)|GPU use' | tr '\n' ';'; echo
  sleep 0.7
done
wait $pid
tail -c 300 gpurun_out/power_bench.json
rocm-smi --showmaxpower 2>/dev/null | grep -i power
