ls /sys/class/drm/ | head; for d in /sys/class/drm/card*/device/hwmon/hwmon*; do echo $d; ls $d | tr '\n' ' '; echo; for f in power1_average power1_input freq1_input freq1_label freq2_input freq2_label power1_cap; do [ -e $d/$f ] && echo "$f: $(cat $d/$f)"; done; done
cat /sys/class/drm/card*/device/pp_dpm_sclk 2>/dev/null | head -5
rocm-smi --showpower --showclocks 2>&1 | head -30
rocm-smi --showpower --showclocks --json 2>&1 | head -30
