(ns maelstrom.gpu
  "Glue a maintainer would add next to maelstrom.core: runs an ensemble of built-in-node tests on the
  MI355X engine (libmaelsim.so through maelstrom.gpu.Native, integration/jni/maelsim_jni.c) and hands
  each history to the UNCHANGED workload checker, exactly where jepsen.core/run! would
  (core.clj:91-100).  Not executable in the build image (no JVM); shown for the boundary."
  (:require [jepsen [checker :as checker] [history :as h]]
            [maelstrom.core :as core])
  (:import (maelstrom.gpu Native)
           (java.nio ByteBuffer ByteOrder)))

(def workloads {:echo 0 :broadcast 1 :g-set 2 :lin-kv 3 :txn-list-append 4 :pn-counter 5 :g-counter 6 :unique-ids 7
                :txn-rw-register 8})
(def node-programs {"builtin:echo" 0 "builtin:broadcast-ff" 1 "builtin:broadcast-ff-echoback" 2
                    "builtin:broadcast-ack-retry" 3 "builtin:broadcast-rpc-all" 4 "builtin:g-set" 5
                    "builtin:raft" 6 "builtin:single-key-txn" 7 "builtin:pn-counter" 8 "builtin:flake-ids" 9 "builtin:lin-kv-proxy" 10
                    "builtin:txn-rw-register-hat" 11})
(def topologies {:grid 0 :line 1 :total 2 :tree 3 :tree2 3 :tree3 4 :tree4 5})
(def fs [:echo :broadcast :read :add :start-partition :stop-partition])
(def types [:invoke :ok :fail :info])

(defn opts->fields
  "core.clj:136-229 option map -> the 16 leading u32 fields of msim_config."
  [{:keys [workload bin nodes concurrency rate time-limit latency nemesis nemesis-interval topology]}]
  (int-array [0 0 (workloads workload) (node-programs bin) (count nodes) (or concurrency (count nodes))
              (long (* 1000 rate)) (* 1000 time-limit) (:mean latency) ({:constant 0 :uniform 1 :exponential 2} (:dist latency))
              0 (topologies topology) (if (:partition nemesis) 1 0) (long (* 1000 nemesis-interval)) 5000 10000]))

(defn decode-op
  "16-byte row -> Jepsen op map (SURVEY.md §8b history surface)."
  [^ByteBuffer rows ^ByteBuffer payload i]
  (let [tl (.getLong rows (* 16 i)) packed (.getInt rows (+ 8 (* 16 i))) value (.getInt rows (+ 12 (* 16 i)))
        len (bit-and (unsigned-bit-shift-right tl 48) 0xffff)
        f (fs (bit-and (bit-shift-right packed 2) 31))
        process (unsigned-bit-shift-right packed 12)]
    (cond-> {:index i :time (bit-and tl 0xffffffffffff) :type (types (bit-and packed 3)) :f f
             :process (if (= process 0xfffff) :nemesis process)
             :value (if (and (= f :read) (pos? len))
                      (vec (for [w (range len) b (range 32)
                                 :when (bit-test (.getInt payload (* 4 (+ value w))) b)] (+ (* 32 w) b)))
                      (when (not= value -1) value))}
      (= 1 (bit-and (bit-shift-right packed 7) 15)) (assoc :error :net-timeout)
      (bit-test packed 11) (assoc :final? true))))

(defn run-ensemble
  "Runs n seeded instances of the test described by CLI `opts`; returns one checker result per instance."
  [opts seed n]
  (let [test (core/maelstrom-test opts)
        ctx  (Native/create (opts->fields opts) seed 0)]
    (try
      (Native/run ctx 0 n)
      (vec (for [i (range n)]
             (let [[^ByteBuffer rows ^ByteBuffer payload] (Native/history ctx i)
                   _ (.order rows ByteOrder/LITTLE_ENDIAN) _ (.order payload ByteOrder/LITTLE_ENDIAN)
                   history (h/history (mapv #(decode-op rows payload %) (range (quot (.capacity rows) 16))))]
               (checker/check (:checker test) test history {}))))
      (finally (Native/destroy ctx)))))
