#!/bin/bash
# Run HERE (not on the GPU box) after `gpurun -- bash scripts/collect_round.sh <tag>`: copies the merged evidence set gpurun_out/<tag>/ into
# profiles/<tag>/ and the four counter summaries into profiles/latest_*_pmc_summary.json (what bench.py opens for roofline.traffic).
#   bash scripts/install_profiles.sh r06j_final [previous dir under profiles/ to replace]
set -e
cd "$(dirname "$0")/.."
TAG=$1; PREV=${2:-}
[ -d gpurun_out/$TAG ] || { echo "gpurun_out/$TAG missing"; exit 1; }
if [ -n "$PREV" ] && [ -d profiles/$PREV ] && [ "$PREV" != "$TAG" ]; then git mv profiles/$PREV profiles/$TAG; fi
mkdir -p profiles/$TAG
cp gpurun_out/$TAG/* profiles/$TAG/
for w in bench shard mult d256; do cp gpurun_out/$TAG/latest_${w}_pmc_summary.json profiles/latest_${w}_pmc_summary.json; done
python3 - <<PY
import json, importlib.util
spec = importlib.util.spec_from_file_location("b", "bench.py"); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
tag = b.kernel_source_tag()
for w in ("bench", "shard", "mult", "d256"):
    t = json.load(open(f"profiles/latest_{w}_pmc_summary.json"))["_meta"]["kernel_source_tag"]
    print(f"latest_{w}_pmc_summary.json: collected on {t}; sources here {tag}: {'current' if t == tag else 'STALE'}")
PY
