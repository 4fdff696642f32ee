set -e
cd /root/repo
for env in "RT_AMD_X=1" "RT_AMD_NO_PRIO=1" "RT_AMD_OVL_MAIN=4 RT_AMD_OVL_MAIN2=4" "RT_AMD_NO_PRIO=1 RT_AMD_OVL_MAIN=4 RT_AMD_OVL_MAIN2=4" "RT_AMD_OVL_SIDE=2" "RT_AMD_NO_PIPELINE=1"; do
  for args in "--width 3840 --height 2160 --world 8 --rank 0" "--width 1920 --height 1080" "--width 3840 --height 2160"; do
    echo "== $env $args"
    env $env timeout -k 10 200 python tools/prof_frame.py --frames 12 --no-profile $args 2>&1 | grep ms_total
  done
done
