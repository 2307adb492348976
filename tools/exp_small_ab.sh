#!/bin/bash
# Interleaved A/B of product libraries on the small-launch jobs of tools/exp_small.py:
#   tools/exp_small_ab.sh <outdir> <rounds> <name>...     (tools/exp/librt_<name>.so; "hip" = the in-tree build)
out=$1; rounds=$2; shift 2
mkdir -p "$out"
for r in $(seq 1 "$rounds"); do
  for v in "$@"; do
    if [ "$v" = hip ]; then lib=$PWD/raytracing_c_amd/librt_hip.so; else lib=$PWD/tools/exp/librt_$v.so; fi
    RT_LIB_PATH=$lib timeout -k 10 300 python tools/exp_small.py > "$out/ab_${v}_$r.jsonl" 2> "$out/ab_${v}_$r.err" || { tail -5 "$out/ab_${v}_$r.err"; exit 1; }
  done
done
python - "$out" "$rounds" "$@" <<'PY'
import json, sys
out, rounds, names = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
rows = {}
for v in names:
    for r in range(1, rounds + 1):
        for l in open(f"{out}/ab_{v}_{r}.jsonl"):
            o = json.loads(l)
            rows.setdefault(o["job"], {}).setdefault(v, []).append((o["ms_first"], o["ms_ordered"]))
print("| job | " + " | ".join(f"{v}: first / ordered ms" for v in names) + " |")
print("|---|" + "---|" * len(names))
for job, d in rows.items():
    print(f"| {job} | " + " | ".join(" ; ".join(f"{a:.3f} / {b:.3f}" for a, b in d.get(v, [])) for v in names) + " |")
PY
