#!/bin/bash
# A/B of decode variants: tools/dec_ab.sh 2 4 ...  (libraries built by `make -C ravvent-basecaller_amd/csrc decvar V=n`)
cd "$(dirname "$0")/.."
python tools/dec_ab.py "$EXTRA" || exit 1
for v in "$@"; do
  RAVVENT_HIP_LIB=$PWD/ravvent-basecaller_amd/csrc/libravvent_hip_cp$v.so python tools/dec_ab.py "$EXTRA" || exit 1
done
