// layout_thresholds.h — the launch sizes from which the eight-clusters-per-wavefront layouts are taken instead of one cluster per
// wavefront (msim_launch_*8 return MSIM_LAYOUT_DOES_NOT_FIT below them and msim_run falls back; MSIM_DEV_FLAGS bit 10 forces the
// packed layout whatever the launch).  The packed kernels are latency-bound — a wavefront's run takes a fixed time whatever the batch —
// so they win from the batch size on where the one-cluster kernels have filled the chip.  Each number is a crossover MEASURED on one
// MI355X (256 CUs); on another part they are the first thing to re-measure.
//
//   constant                     value   crossover measured                                                      profile
//   MSIM_UID8_MIN_CLUSTERS        4096   unique-ids 3 nodes: 18.9 ms packed against 38.8 ms at 4096 clusters      profiles/r03am_uid8.txt
//   MSIM_CRDT8_MIN_CLUSTERS       4096   pn-counter 5 nodes: 7.0 against 30.4 ms at 16384; even at 4096           profiles/r03an_crdt8.txt
//   MSIM_BCAST8_MIN_CLUSTERS     12288   broadcast 5 nodes: 23.7 against 14.8 ms at 4096 (loses), 26.8 / 47.6 at 16384   profiles/r03ar_bcast8.txt
//   MSIM_HAT8_MIN_CLUSTERS_PER_NODE 3200 txn-rw-register: 2 nodes even near 6400, 3 nodes 8192..16384, 5 nodes near 16384   profiles/r03ae_hat8.txt
//   MSIM_KAFKA8_MIN_CLUSTERS      8192   kafka 1 / 3 / 5 / 7 nodes: packed 30 / 73 / 88 / 93 ms against 21 / 50 / 61 / 63 at 4096 (loses), 31 / 77 / 92 / 97 against 34 / 80 / 94 / 97 at 8192 (even), 34 / 83 / 99 / 105 against 65 / 151 / 177 / 184 at 16384   profiles/r05_kafka8_threshold_sweep.jsonl
//   MSIM_DT8_MIN_CLUSTERS        12288   txn-list-append over the Datomic-style node, 2 / 5 nodes: packed 125 / 294 ms against 56 / 142 at 4096 (loses), 136 / 326 against 108 / 277 at 8192, 158 / 383 against 207 / 536 at 16384, 316 / 760 against 401 / 1042 at 32768   profiles/r05_dt8_threshold_sweep.jsonl
//   MSIM_SVC4_MIN_CLUSTERS        4096   lin-kv proxy 5 nodes c=10: four per wavefront 16.0 against 17.2 ms at 4096, 18.0 / 31.9 at 8192, 24.9 / 59.6 at 16384 (loses below: 15.6 / 12.3 at 2048)   profiles/r06f_svc4_threshold_sweep.jsonl
//   MSIM_TXNG4_MIN_CLUSTERS       8192   txn-list-append, single-root node, 1 node x 10 workers: four per wavefront 107.6 against 116.6 ms at 8192, 134.5 / 222.6 at 16384 (loses below: 98 / 61 at 4096); 5 nodes x 2: 105 / 176 at 8192, 130 / 323 at 16384   profiles/r06f_txng4_threshold_sweep.jsonl
//   MSIM_DTG4_MIN_CLUSTERS       16384   txn-list-append, Datomic-style node, 1 node x 10 workers at latency 0 (the reference's invocation): four per wavefront 708.5 against 952.9 ms at 16384 (loses below: 549 / 493 at 8192); latency 5: 97.7 / 131.1; 2 nodes x 6 + partitions: 315 / 357   profiles/r06f_dtg4_threshold_sweep.jsonl
#ifndef MSIM_LAYOUT_THRESHOLDS_H
#define MSIM_LAYOUT_THRESHOLDS_H
#define MSIM_UID8_MIN_CLUSTERS 4096u
#define MSIM_CRDT8_MIN_CLUSTERS 4096u
#define MSIM_BCAST8_MIN_CLUSTERS 12288u
#define MSIM_HAT8_MIN_CLUSTERS_PER_NODE 3200u
#define MSIM_KAFKA8_MIN_CLUSTERS 8192u
#define MSIM_DT8_MIN_CLUSTERS 12288u
#define MSIM_SVC4_MIN_CLUSTERS 4096u
#define MSIM_TXNG4_MIN_CLUSTERS 8192u
#define MSIM_DTG4_MIN_CLUSTERS 16384u
#endif
