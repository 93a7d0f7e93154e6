# Developer tool (GPU box): BASELINE cfg3 and its lossy variants with the nodes' sets in HBM scratch (flags 0: the default since round 4) and in LDS (MSIM_DEV_FLAGS bit 14 = 0x4000)
for fl in 0 0x4000; do
  echo "== flags=$fl"
  MSIM_DEV_FLAGS=$fl python tools/bench_configs.py "cfg3 g-set n=100 lat100 exponential" "cfg3 g-set n=100 lat100 exponential p_loss 0.05" "cfg3 g-set n=100 lat100 exponential p_loss 0.5" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['config'], 'sim_ms', round(d['sim_ms'], 1), 'valid', d['valid'], 'flagged', d['flagged'])"
done
