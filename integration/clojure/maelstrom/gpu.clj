(ns maelstrom.gpu
  "Glue a maintainer would add next to maelstrom.core: runs an ensemble of built-in-node tests on the
  MI355X engine (libmaelsim.so through maelstrom.gpu.Native, integration/jni/maelsim_jni.c) and hands
  each history to the UNCHANGED workload checker, exactly where jepsen.core/run! would
  (core.clj:91-100).  Not executable in the build image (no JVM); shown for the boundary.

  The history crosses as history.edn text (Native/historyEdn = msim_history_edn_rows): one op map per line with the
  reference's :value shapes for EVERY workload (echo maps, read vectors, lin-kv independent tuples, txn micro-op
  vectors, counter longs, flake ids, nemesis grudges), so this side needs no per-workload row decoder."
  (:require [clojure.edn :as edn]
            [clojure.string :as str]
            [jepsen [checker :as checker] [history :as h]]
            [maelstrom.core :as core])
  (:import (maelstrom.gpu Native)
           (java.nio ByteBuffer ByteOrder)))

(def workloads {:echo 0 :broadcast 1 :g-set 2 :lin-kv 3 :txn-list-append 4 :pn-counter 5 :g-counter 6 :unique-ids 7
                :txn-rw-register 8})
(def node-programs {"builtin:echo" 0 "builtin:broadcast-ff" 1 "builtin:broadcast-ff-echoback" 2
                    "builtin:broadcast-ack-retry" 3 "builtin:broadcast-rpc-all" 4 "builtin:g-set" 5
                    "builtin:raft" 6 "builtin:single-key-txn" 7 "builtin:pn-counter" 8 "builtin:flake-ids" 9 "builtin:lin-kv-proxy" 10
                    "builtin:txn-rw-register-hat" 11 "builtin:multi-key-txn" 12 "builtin:tso-ids" 13 "builtin:kafka" 14
                    ; demo/ruby/datomic_list_append.rb, what core.clj:113-114 runs for txn-list-append (MSIM_NODE_TXN_DATOMIC)
                    "builtin:datomic" 15})
(def topologies {:grid 0 :line 1 :total 2 :tree 3 :tree2 3 :tree3 4 :tree4 5})
(def latency-dists {:constant 0 :uniform 1 :exponential 2})
(def services {"lin-kv" 0 "seq-kv" 1 "lww-kv" 2})
(def consistency-models {:strict-serializable 0 :serializable 1 :snapshot-isolation 2 :read-committed 3 :read-uncommitted 4})

; byte offsets of the msim_config fields (include/maelsim.h; checked against the header by tests/test_jni_stub.py)
(def config-size 120)
(def config-offsets
  {:struct-size 0 :abi-version 4 :workload 8 :node-program 12 :n-nodes 16 :concurrency 20 :rate-mhz 24 :time-limit-ms 28
   :latency-mean-ms 32 :latency-dist 36 :p-loss-q32 40 :topology 44 :nemesis-mask 48 :nemesis-interval-ms 52
   :client-timeout-ms 56 :quiesce-ms 60 :seed 64 :max-values 72 :max-rows 76 :max-payload-words 80 :inbox-capacity 84
   :spill-capacity 88 :journal-capacity 92 :key-count 96 :max-txn-length 100 :max-writes-per-key 104 :proxy-service 108
   :consistency-model 112 :replication-words 116})

(defn opts->config
  "core.clj:136-229 option map -> msim_config bytes: the engine's defaults for the workload (Native/configDefaults =
  msim_config_defaults), then every option the CLI carries."
  ^bytes [{:keys [workload bin nodes concurrency rate time-limit latency nemesis nemesis-interval topology
                  key-count max-txn-length max-writes-per-key consistency-models service journal-capacity]} seed]
  (let [bytes (Native/configDefaults (workloads workload) (count nodes))
        buf   (doto (ByteBuffer/wrap bytes) (.order ByteOrder/LITTLE_ENDIAN))
        put!  (fn [k v] (when (some? v) (.putInt buf (config-offsets k) (unchecked-int v))))]
    (assert (= config-size (alength bytes)))
    (put! :node-program (node-programs bin))
    (put! :concurrency concurrency)
    (put! :rate-mhz (long (* 1000 rate)))
    (put! :time-limit-ms (* 1000 time-limit))
    (put! :latency-mean-ms (:mean latency))
    (put! :latency-dist (latency-dists (:dist latency)))
    (put! :topology (topologies topology))
    (put! :nemesis-mask (if (:partition nemesis) 1 0))
    (put! :nemesis-interval-ms (long (* 1000 nemesis-interval)))
    (put! :key-count key-count)
    (put! :max-txn-length max-txn-length)
    (put! :max-writes-per-key max-writes-per-key)
    (put! :consistency-model (some-> consistency-models first consistency-models))
    (put! :proxy-service (services service))
    (put! :journal-capacity journal-capacity)
    (.putLong buf (config-offsets :seed) seed)
    bytes))

(defn parse-history
  "history.edn text (one op map per line) -> jepsen.history"
  [^String text]
  (->> (str/split-lines text)
       (remove str/blank?)
       (mapv edn/read-string)
       h/history))

(defn run-ensemble
  "Runs n seeded instances of the test described by CLI `opts`; returns one checker result per instance."
  [opts seed n]
  (let [test (core/maelstrom-test opts)
        ctx  (Native/create (opts->config opts seed) 0)]
    (try
      (Native/run ctx 0 n)
      (vec (for [i (range n)]
             (let [history (parse-history (Native/historyEdn ctx i))]
               (checker/check (:checker test) test history {}))))
      (finally (Native/destroy ctx)))))
