cd $GRAFT_REPO_ROOT
ls /sys/class/drm/ | head; for c in /sys/class/drm/card*/device; do echo "== $c"; ls $c | grep -E "pp_dpm|hwmon|gpu_busy|current" | head; for f in pp_dpm_sclk pp_dpm_mclk gpu_busy_percent; do [ -f $c/$f ] && { echo "-- $f"; cat $c/$f | head -12; }; done; for h in $c/hwmon/hwmon*; do ls $h | head -30 | tr '\n' ' '; echo; for f in freq1_input freq1_label freq2_input freq2_label power1_average power1_input power1_label; do [ -f $h/$f ] && echo "$f: $(cat $h/$f)"; done; done; done 2>/dev/null | head -80
python scripts/sustained_load.py 6 > /dev/null 2>&1 &
sleep 3
for i in 1 2 3; do for c in /sys/class/drm/card*/device; do for h in $c/hwmon/hwmon*; do echo "load: freq1 $(cat $h/freq1_input 2>/dev/null) power $(cat $h/power1_average 2>/dev/null || cat $h/power1_input 2>/dev/null)"; done; [ -f $c/pp_dpm_sclk ] && grep '\*' $c/pp_dpm_sclk; done; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | head -4; sleep 0.5; done
wait
