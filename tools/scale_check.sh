#!/bin/bash
# First contact with a multi-GPU node as ONE command (BASELINE.json configs[4]: MAX_ADDR = 2^21, rows sharded over the GPUs).
# The last column compares with the one-GPU prediction of tools/scaling_model.py (profiles/r06_scaling_model.json; rehearsals on one
# GPU time-share it, so the ratio means something on a multi-GPU node only).
# For N in 1 2 4 8 (or $NS): the committed 2^21 digests through the native group (fheram_group_*, one process) and through
# one process per GPU over RCCL, then bench.py in both modes; prints a table.  Nothing here needs /root/reference.
#   tools/scale_check.sh                      # on an 8-GPU node
#   REHEARSE=1 tools/scale_check.sh           # on ONE GPU: every rank / shard on device 0, gloo instead of RCCL (what -m gpu runs)
#   NS="1 2" LOG=18 STEPS=5 tools/scale_check.sh
set -u
cd "$(dirname "$0")/.."
NS=${NS:-"1 2 4 8"}
LOG=${LOG:-21}
STEPS=${STEPS:-10}
WARM=${WARM:-3}
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
EXTRA=""; BACKEND=nccl
if [ "${REHEARSE:-0}" = "1" ]; then EXTRA="--all-ranks-device0"; BACKEND=gloo; fi
PORT=29580
RANKS_MAX=${RANKS_MAX:-64}     # the one-process-per-GPU legs are skipped above this N (a 1-GPU rehearsal box admits 6 GPU processes)
fail=0
out=$(mktemp -d)
printf "%-3s %-22s %-10s %-12s %-10s %-10s %-10s %s\n" N mode digests "RAM ops/s" read_ms rpw_ms write_ms "measured / predicted (profiles/r06_scaling_model.json)"
for n in $NS; do
  # --- one process, n devices (native group)
  python tests/scale_digest_worker.py --mode group --n $n --log-max-addr $LOG $EXTRA > $out/gd_$n.json 2> $out/gd_$n.err; gd=$?
  python bench.py --gpus $n --mode group --total-log-max-addr $LOG --steps $STEPS --warmup $WARM $EXTRA > $out/gb_$n.json 2> $out/gb_$n.err; gb=$?
  python - "$out/gb_$n.json" $n group $gd $gb $LOG <<'PY'
import json, sys
f, n, mode, gd, gb, log = sys.argv[1:7]
def predicted(n, log):
    try:
        import os
        m = json.load(open(os.path.join("profiles", "r06_scaling_model.json")))
        for r in m["rows"]:
            if r["mode"] == "strong" and r["total_log_max_addr"] == int(log) and r["gpus"] == int(n):
                return r["ram_ops_s"]
    except Exception:
        pass
    return None

try: d = json.loads(open(f).read().strip().split("\n")[-1])
except Exception: d = {}
pv = predicted(n, log)
print("%-3s %-22s %-10s %-12s %-10s %-10s %-10s %s" % (n, "group (1 process)", "ok" if gd == "0" else "FAILED",
      ("%.1f" % d["value"]) if "value" in d and gb == "0" else "FAILED", *[("%.3f" % d[k]) if k in d else "-" for k in ("read_ms", "read_prepare_write_ms", "write_ms")],
      ("%.2f (predicted %.1f)" % (d["value"] / pv, pv)) if (pv and "value" in d) else "-"))
PY
  [ $gd -ne 0 ] || [ $gb -ne 0 ] && fail=1
  # --- one process per GPU over RCCL (gloo in the rehearsal)
  if [ $n -gt $RANKS_MAX ]; then continue; fi
  PORT=$((PORT + 1))
  python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $PORT \
      tests/scale_digest_worker.py --mode ranks --log-max-addr $LOG --dist-backend $BACKEND $EXTRA > $out/rd_$n.json 2> $out/rd_$n.err; rd=$?
  PORT=$((PORT + 1))
  if [ $n -eq 1 ]; then
    python bench.py --gpus 1 --log-max-addr $LOG --steps $STEPS --warmup $WARM --no-cpu-baseline > $out/rb_$n.json 2> $out/rb_$n.err; rb=$?
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $PORT \
        bench.py --gpus $n --total-log-max-addr $LOG --steps $STEPS --warmup $WARM --dist-backend $BACKEND $EXTRA > $out/rb_$n.json 2> $out/rb_$n.err; rb=$?
  fi
  python - "$out/rb_$n.json" $n ranks $rd $rb $BACKEND $LOG <<'PY'
import json, sys
f, n, mode, rd, rb, be, log = sys.argv[1:8]
def predicted(n, log):
    try:
        import os
        m = json.load(open(os.path.join("profiles", "r06_scaling_model.json")))
        for r in m["rows"]:
            if r["mode"] == "strong" and r["total_log_max_addr"] == int(log) and r["gpus"] == int(n):
                return r["ram_ops_s"]
    except Exception:
        pass
    return None

try: d = json.loads([l for l in open(f).read().strip().split("\n") if l.startswith("{")][-1])
except Exception: d = {}
pv = predicted(n, log)
print("%-3s %-22s %-10s %-12s %-10s %-10s %-10s %s" % (n, "1 process/GPU (%s)" % ("RCCL" if be == "nccl" else be), "ok" if rd == "0" else "FAILED",
      ("%.1f" % d["value"]) if "value" in d and rb == "0" else "FAILED", *[("%.3f" % d[k]) if k in d else "-" for k in ("read_ms", "read_prepare_write_ms", "write_ms")],
      ("%.2f (predicted %.1f)" % (d["value"] / pv, pv)) if (pv and "value" in d) else "-"))
PY
  [ $rd -ne 0 ] || [ $rb -ne 0 ] && fail=1
done
if [ $fail -ne 0 ]; then echo "scale_check: FAILED (logs in $out)"; tail -n 5 $out/*.err 2>/dev/null | tail -n 40; exit 1; fi
echo "scale_check: every digest reproduced at every N (logs in $out)"
