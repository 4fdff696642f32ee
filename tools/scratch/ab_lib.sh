#!/bin/bash
# A/B of two builds of the library: $1 = path of the other build
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
other=$PWD/$1
for args in "--width 1920 --height 1080" "--width 3840 --height 2160 --frames 30" "--width 3840 --height 2160 --world 8 --rank 0" "--workload pool --sh-rays 1" "--workload pool" "--workload original --width 500 --height 500" "--workload synth90k --frames 20" "--workload synth1M --frames 10"; do
  for lib in head other; do
    if [ $lib = other ]; then export RT_AMD_LIB=$other; else unset RT_AMD_LIB; fi
    echo -n "$lib "; timeout -k 10 200 python tools/b2b.py $args 2>&1 | grep "^b2b" | cut -c1-110 || exit 1
  done
done
