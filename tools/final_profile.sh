#!/bin/bash
# The round's final records, each step under its own time limit:
#   tools/final_profile.sh passes   counter passes of the headline product (then tools/update_traffic.py)
#   tools/final_profile.sh epochs   kernel traces of the two forms of the training step
#   tools/final_profile.sh bench    kernel trace of the default bench command
set -u
export TMPDIR=/tmp
out=gpurun_out/r3f
mkdir -p $out
case "${1:-all}" in
passes)
  bash tools/profile_passes.sh $out/final_f256 tools/orderings.py --scale 24 --features 256 --order scrambled > $out/final_f256.log 2>&1 || exit 1
  ;;
epochs)
  for form in reference restructured; do
    timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ep_$form/trace -- python3 tools/epoch_probe.py --form $form --steps 6 > $out/ep_$form.log 2>&1 || exit 1
    python3 tools/summarize_profile.py $out/ep_$form $out/ep_$form.md > /dev/null || exit 1
  done
  ;;
bench)
  timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench/trace -- python3 bench.py > $out/bench.json 2> $out/bench.err || exit 1
  python3 tools/summarize_profile.py $out/bench $out/bench.md > /dev/null || exit 1
  ;;
*)
  echo "usage: $0 passes|epochs|bench"; exit 2
  ;;
esac
echo "$1 done"
