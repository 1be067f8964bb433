#!/bin/bash
# build glimpse_amd/lib/base.so from HEAD (for tools/ab.sh), then rebuild the working tree
set -e
cd "$(dirname "$0")/.."
dirty=$(git status --porcelain --untracked-files=no | wc -l)
[ "$dirty" -gt 0 ] && git stash -q
python -m glimpse_amd.build >/dev/null 2>&1 || { [ "$dirty" -gt 0 ] && git stash pop -q; exit 1; }
cp glimpse_amd/lib/libglimpse_hip.so glimpse_amd/lib/base.so
[ "$dirty" -gt 0 ] && git stash pop -q
python -m glimpse_amd.build >/dev/null 2>&1
ls -la glimpse_amd/lib/
