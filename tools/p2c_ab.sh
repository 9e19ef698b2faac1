#!/bin/bash
# chain kernel A/B on ONE box, alternating: the product library vs reference builds under tools/*.bin (E2EMV_LIBRARY); per forward
# of configs[1]: us per chain launch, attention launch, first-layer q|k|v launch (tools/p2c_stamps.py --one)
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2 3; do
  for lib in "$@"; do
    if [ "$lib" = product ]; then echo -n "product      "; python tools/p2c_stamps.py --one 2>&1 | grep dbg=
    else echo -n "$lib "; E2EMV_LIBRARY=$GRAFT_REPO_ROOT/tools/$lib python tools/p2c_stamps.py --one 2>&1 | grep dbg=; fi
  done
done
