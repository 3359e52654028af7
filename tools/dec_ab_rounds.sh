#!/bin/bash
# A/B of decode variants with the launch-to-launch and process-to-process scatter averaged out: `rounds` passes over the default library and
# the variants libravvent_hip_cp<v>.so (make -C ravvent-basecaller_amd/csrc decvar V=<v> DECFLAGS=...), round-robin.
# usage: tools/dec_ab_rounds.sh <rounds> <v> [<v> ...]
cd "$(dirname "$0")/.."
rounds=$1; shift
for r in $(seq 1 $rounds); do
  python tools/dec_ab.py || exit 1
  for v in "$@"; do
    RAVVENT_HIP_LIB=$PWD/ravvent-basecaller_amd/csrc/libravvent_hip_cp$v.so python tools/dec_ab.py || exit 1
  done
done
