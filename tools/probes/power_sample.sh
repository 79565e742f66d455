# package power and shader clock sampled every 50 ms while the bench step runs (is the step power-bound?)
cd $GRAFT_REPO_ROOT
hw=$(ls -d /sys/class/drm/card*/device/hwmon/hwmon* 2>/dev/null | head -1)
echo "hwmon: $hw"; ls $hw 2>/dev/null | tr '\n' ' '; echo
cat $hw/power1_cap 2>/dev/null; cat $hw/power1_cap_max 2>/dev/null
python3 bench.py --no-cpu-baseline --no-ar --no-extra --steps 60 > gpurun_out/pw_line.json 2>/dev/null &
pid=$!
for i in $(seq 1 700); do
  p=$(cat $hw/power1_average 2>/dev/null || cat $hw/power1_input 2>/dev/null); f=$(cat $hw/freq1_input 2>/dev/null); t=$(cat $hw/temp1_input 2>/dev/null)
  echo "$i $p $f $t"
  sleep 0.05
done > gpurun_out/pw_samples.txt
wait $pid
awk '$2 > 600000000 {p+=$2; f+=$3; n++} END {print "busy samples (> 600 W):", n, "mean power W", p/n/1e6, "mean sclk MHz", f/n/1e6}' gpurun_out/pw_samples.txt
awk '{print int($2/1e8)*100}' gpurun_out/pw_samples.txt | sort -n | uniq -c
awk '$2 > 600000000 {print int($3/1e8)*100}' gpurun_out/pw_samples.txt | sort -n | uniq -c
