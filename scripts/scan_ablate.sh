#!/bin/bash
# GPU-box helper (diagnostic): rebuild the library with each flag set (FORCED: build.py only rebuilds on newer sources) and time the scan probe
cd ${GRAFT_REPO_ROOT:-.}
for f in "$@"; do
  SS_EXTRA_HIPCC_FLAGS="$f" python -c "import sys; sys.path.insert(0, '.'); from scenesplat_amd import build; build.build(force=True, verbose=False)" > /dev/null 2>&1
  SS_EXTRA_HIPCC_FLAGS="$f" python scripts/scan_probe.py | grep scan
done
