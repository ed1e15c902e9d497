(ns tigerbeetle.checker.gpu-linear
  "jepsen.checker/linearizable, answered by libtbcheck.so on an MI355X (hand-written HIP behind a C-ABI,
   include/tbcheck.h).  Falls back to stock Knossos when the library, a GPU, or a device model for the
   caller's knossos model is missing.

   NEVER EXECUTED where it was written (no JVM there): it is written to the header, and what a machine
   can check without a JVM is checked by tests/test_clj_shim.py -- the `abi` table below against the
   ctypes layouts (which tests/test_abi.py pins to the header with gcc), every byte offset used in this file
   against that table, the enum codes against the header, and bracket balance.

   Reference call sites this namespace is for (the reference never calls the search today):
   src/tigerbeetle/workloads/set_full.clj:155-158, src/tigerbeetle/tests/ledger.clj:363-367
   (clj/patches/*.patch)."
  (:require [jepsen.checker :as checker]
            [knossos.history :as history]
            [knossos.model :as model]
            [knossos.op :as op]
            [clojure.tools.logging :refer [warn]])
  (:import (com.sun.jna Callback Function Memory NativeLibrary Pointer)
           (com.sun.jna.ptr IntByReference PointerByReference)))

;; ---------------------------------------------------------------------------------------------------------
;; The C-ABI as this file uses it: struct sizes and field offsets of TBC_ABI_VERSION 2 (LP64), enum codes.
;; tests/test_clj_shim.py reads this form as EDN and compares every number with the library's ctypes binding.
;; ---------------------------------------------------------------------------------------------------------
(def abi
  {:version 2
   :events       {:size 48  :n 0 :type 8 :process 16 :f 24 :a 32 :b 40}
   :ops          {:size 72  :n 0 :n_events 4 :f 8 :a 16 :b 24 :process 32 :inv_pos 40 :ret_pos 48 :pool 56
                  :pool_len 64 :n_process 68}
   :model        {:size 32  :kind 0 :init 4 :table 8 :n_states 16 :n_classes 20 :n_keys 24 :flags 28}
   :opts         {:size 64  :algorithm 0 :device 4 :time_limit_ms 8 :max_steps 16 :max_visited_bytes 24
                  :want_witness 32 :visited_per_op 36 :search_width 40 :round_budget 44 :lookahead 48
                  :dominance 52 :lanes_per_history 56 :list_order 60}
   :config       {:size 84  :state 0 :last_op 4 :n_pending 8 :n_linearized 12 :pending 16 :linearized_mask 80}
   :result       {:size 960 :valid 0 :cause 4 :analyzer 8 :fail_op 12 :prev_ok_op 16 :final_state 20
                  :n_witness 24 :search_width 28 :witness 32 :n_configs 40 :configs 44}
   :batch_desc   {:size 112 :n_hist 0 :op_off 8 :n_events 16 :n_process 24 :cols 32 :model_aux 104}
   :setfull_in   {:size 56  :n_elements 0 :n_reads 4 :words_per_row 8 :device 12 :add_invoke 16 :add_ok 24
                  :read_invoke 32 :read_ok 40 :present 48}
   :setfull_out  {:size 48  :known 0 :last_present 8 :last_absent 16 :ns_scan 24 :bytes_scanned 32
                  :bytes_matrix 40}
   :setfull_rows {:size 72  :n_elements 0 :n_reads 4 :device 8 :reserved0 12 :add_invoke 16 :add_ok 24
                  :read_invoke 32 :read_ok 40 :top 48 :exc_off 56 :exc 64}
   ;; streaming (round 6): pointers into one pinned slot of a batch; what the last consumed input cost
   :batch_input  {:size 64  :n_hist_cap 0 :reserved0 4 :ops_cap 8 :op_off 16 :n_events 24 :n_process 32 :word 40
                  :inv_pos 48 :ret_pos 56}
   :input_info   {:size 56  :n_hist 0 :pending 4 :total_ops 8 :bytes_copied 16 :ns_copy 24 :inputs_consumed 32
                  :lists_regrown 40 :n_hist_cap 44 :ops_cap 48}
   ;; a run that is out, seen from another thread (tbc_batch_progress)
   :progress     {:size 24  :n_histories 0 :n_decided 4 :phase 8 :running 12 :elapsed_ns 16}
   :enums        {:type     {:invoke 0 :ok 1 :fail 2 :info 3}
                  :f        {:read 0 :write 1 :cas 2 :acquire 3 :release 4 :add 5 :txn 6 :transfer 7 :class 8}
                  :model    {:register 0 :cas-register 1 :mutex 2 :table 3 :multi-register 4 :set 5 :bank 6}
                  :alg      {:competition 0 :wgl 1 :linear 2}
                  :valid    {:valid 1 :invalid 0 :unknown -1}
                  :cause    {:none 0 :time-limit 1 :step-limit 2 :memory 3}
                  :max-final-configs 10
                  :config-pending 16
                  :wire-nil 255                  ; TBC_WIRE_NIL
                  :comm-id-bytes 128}})          ; TBC_COMM_ID_BYTES

(defn- o
  "Byte offset of field k of struct s (or its :size)."
  ^long [s k]
  (long (get-in abi [s k])))

(defn- struct ^Memory [s]
  (doto (Memory. (o s :size)) (.clear)))

(def ^:private lib
  (delay (try (let [l (NativeLibrary/getInstance "tbcheck")]
                (if (= (:version abi) (.invokeInt (.getFunction l "tbc_version") (object-array 0)))
                  l
                  (do (warn "libtbcheck has another ABI version than this shim; using stock Knossos") nil)))
              (catch Throwable _ nil))))

(defn- f ^Function [name] (.getFunction ^NativeLibrary @lib name))

(def type-code (get-in abi [:enums :type]))
(def f-code    (get-in abi [:enums :f]))
(def NIL Integer/MIN_VALUE)                     ; TBC_NIL
(def NO-OP (unchecked-int 0xFFFFFFFF))          ; TBC_NO_OP / TBC_POS_CRASHED

;; ---------------------------------------------------------------------------------------------------------
;; models
;; ---------------------------------------------------------------------------------------------------------
(defrecord Bank [balances negative-balances?]
  ;; Knossos ships no bank model; this is the one DESIGN.md section 1 specifies from tests/ledger.clj:89-114:
  ;; :transfer moves :amount from :debit-acct to :credit-acct (balance = credits - debits), :read is consistent iff
  ;; its {acct balance} map equals the state (a nil read always is, as everywhere in Knossos).
  model/Model
  (step [this op]
    (let [v (:value op)]
      (case (:f op)
        :transfer (let [{:keys [debit-acct credit-acct amount]} (if (map? v) v (nth (first v) 2))
                        b (-> balances (update debit-acct - amount) (update credit-acct + amount))]
                    (if (and (not negative-balances?) (some neg? (vals b)))
                      (model/inconsistent (str "negative balance after " v))
                      (assoc this :balances b)))
        :read     (if (or (nil? v) (= v balances))
                    this
                    (model/inconsistent (str "can't read " v " from " balances)))))))

(defn bank-model
  "The bank model over `accounts` (all balances 0), negative balances allowed unless told otherwise
   (core.clj:217-219 :negative-balances?)."
  ([accounts] (bank-model accounts true))
  ([accounts negative-balances?] (->Bank (zipmap accounts (repeat 0)) negative-balances?)))

(defn- model->native
  "[kind init n-keys] for the models the device evaluates directly (tbc_model.kind / .init / .n_keys)."
  [m]
  (let [k (get-in abi [:enums :model])]
    (condp instance? m
      knossos.model.CASRegister [(:cas-register k) (or (:value m) NIL) 0]
      knossos.model.Register    [(:register k) (or (:value m) NIL) 0]
      knossos.model.Mutex       [(:mutex k) (if (:locked? m) 1 0) 0]
      knossos.model.Set         (when (empty? (:s m)) [(:set k) 0 0])
      Bank                      (when (and (:negative-balances? m) (every? zero? (vals (:balances m)))
                                           (<= (count (:balances m)) 16))
                                  [(:bank k) 0 (count (:balances m))])
      nil)))                                     ; anything else: memo table (memo-table below) or Knossos

(defn- state->model
  "tbc_config.state / tbc_result.final_state back to a knossos model of the caller's kind."
  [m st]
  (condp instance? m
    knossos.model.CASRegister (model/cas-register (when (not= st NIL) st))
    knossos.model.Register    (model/register (when (not= st NIL) st))
    knossos.model.Mutex       (if (zero? st) (model/mutex) (model/step (model/mutex) {:f :acquire}))
    m))                                          ; state-free device models (set, bank): configs carry no state

;; ---------------------------------------------------------------------------------------------------------
;; history -> columns
;; ---------------------------------------------------------------------------------------------------------
(defn- int-pool ^Memory [ints]
  (let [m (Memory. (* 4 (max 1 (count ints))))]
    (dorun (map-indexed (fn [i v] (.setInt m (* 4 i) (unchecked-int v))) ints))
    m))

(defn- set-encoding
  "history (client ops of ONE key, :index = row) -> {:a {row -> a-column value} :pool [ints]}.
   Adds are numbered in completion order (crashed ones last, by invocation); pool[0..R] = adds completed before
   each completion rank; an :ok read with value S gets a record {|S| or -1 if S holds an element nobody adds,
   number of leading ones, bitset words over the add numbers}.  Elements must be unique (else: memo table).
   Transliteration of jepsen-tigerbeetle_amd/knossos/_analysis.py::_set_direct (the tested encoder)."
  [hist]
  (let [pairs   (history/pair-index hist)                 ; invoke <-> completion
        adds    (filter #(and (op/invoke? %) (= :add (:f %))) hist)
        done    (fn [x] (let [c (pairs x)] (when (and c (op/ok? c)) c)))
        failed? (fn [x] (some-> (pairs x) op/fail?))
        live    (sort-by (comp :index done) (filter done adds))
        crashed (remove #(or (done %) (failed? %)) adds)
        order   (vec (concat live crashed))
        j-of    (into {} (map-indexed (fn [j x] [(:value x) j]) order))
        nwords  (max 1 (quot (+ (count order) 31) 32))
        oks     (sort-by :index (filter op/ok? hist))
        nb      (reductions + 0 (map #(if (= :add (:f %)) 1 0) oks))   ; nadds_before[0..R]
        pool    (transient (vec nb))
        a-col   (transient {})]
    (doseq [x adds :when (not (failed? x))]
      (assoc! a-col (:index x) (j-of (:value x)))
      (when-let [c (pairs x)] (assoc! a-col (:index c) (j-of (:value x)))))
    (doseq [x hist :when (and (op/ok? x) (= :read (:f x)) (some? (:value x)))]
      (let [v     (set (:value x))
            js    (keep j-of v)
            ok?   (= (count js) (count v))
            bits  (reduce (fn [ws j] (update ws (quot j 32) bit-or (bit-shift-left 1 (rem j 32)))) (vec (repeat nwords 0)) js)
            lead  (count (take-while #(bit-test (bits (quot % 32)) (rem % 32)) (range (count order))))]
        (assoc! a-col (:index x) (count pool))
        (conj! pool (if ok? (count v) -1))
        (conj! pool lead)
        (doseq [w bits] (conj! pool (unchecked-int w)))))
    {:a (persistent! a-col) :pool (persistent! pool)}))

(defn- transfer-map [v] (if (map? v) v (nth (first v) 2)))     ; ledger->bank leaves [[:t id {...}]] (tests/ledger.clj:110)

(defn- bank-encoding
  "ledger->bank ops (tests/ledger.clj:89-114): :transfer {:debit-acct :credit-acct :amount}, :read {acct balance}.
   pool[0 .. (R+1)*A) = balances before each completion rank (transfers of completed calls applied in
   completion order), then {debit idx, credit idx, amount} per transfer and the A balances per :ok read.
   Transliteration of _analysis.py::_bank_direct."
  [hist accounts]
  (let [idx   (zipmap accounts (range))
        A     (count accounts)
        pairs (history/pair-index hist)
        oks   (sort-by :index (filter op/ok? hist))
        step  (fn [bal x] (if (= :transfer (:f x))
                            (let [{:keys [debit-acct credit-acct amount]} (transfer-map (:value (or (pairs x) x)))]
                              (-> bal (update (idx debit-acct) - amount) (update (idx credit-acct) + amount)))
                            bal))
        bals  (reductions step (vec (repeat A 0)) oks)
        pool  (transient (vec (apply concat bals)))
        a-col (transient {})]
    (doseq [x hist :when (op/invoke? x)]
      (let [c (pairs x)]
        (case (:f x)
          :transfer (let [{:keys [debit-acct credit-acct amount]} (transfer-map (:value x)) off (count pool)]
                      (conj! pool (idx debit-acct)) (conj! pool (idx credit-acct)) (conj! pool amount)
                      (assoc! a-col (:index x) off) (when c (assoc! a-col (:index c) off)))
          :read     (when (and c (op/ok? c) (:value c))
                      (assoc! a-col (:index c) (count pool))
                      (doseq [acct accounts] (conj! pool (get (:value c) acct))))
          nil)))
    {:a (persistent! a-col) :pool (persistent! pool)}))

(defn- columns
  "Client ops only -- the reference filters on (int? process) the same way (tests/ledger.clj:94,204,228) --
   re-indexed 0..n-1, as direct buffers for tbc_events.  enc (set / bank) overrides the `a` column per row."
  [hist m]
  (let [ops (vec (map-indexed (fn [i x] (assoc x :index i)) (filter (comp int? :process) hist)))
        n   (count ops)
        enc (condp instance? m
              knossos.model.Set (set-encoding ops)
              Bank              (bank-encoding ops (keys (:balances m)))
              nil)
        typ (Memory. (max 1 n)) prc (Memory. (* 4 (max 1 n)))
        fc  (Memory. (max 1 n)) a   (Memory. (* 4 (max 1 n))) b (Memory. (* 4 (max 1 n)))]
    (dotimes [i n]
      (let [{:keys [type f value process]} (nth ops i)
            [va vb] (cond
                      enc        [(get (:a enc) i) nil]
                      (= f :cas) (or value [nil nil])
                      :else      [value nil])]
        (.setByte typ i (byte (type-code type)))
        (.setInt  prc (* 4 i) (int process))
        (.setByte fc  i (byte (f-code f)))
        (.setInt  a (* 4 i) (int (if (nil? va) NIL va)))
        (.setInt  b (* 4 i) (int (if (nil? vb) NIL vb)))))
    {:ops ops :n n :type typ :process prc :f fc :a a :b b :pool (some-> enc :pool)}))

;; JNA LIFETIMES.  Memory.setPointer writes the pointee's ADDRESS into a struct and keeps no reference to the pointee; a Memory that is
;; only reachable as "the thing whose address I wrote somewhere" is garbage, and its finalizer frees the native block -- while libtbcheck
;; is reading it.  Clojure makes it worse: locals are cleared after their last use, so even a let-bound buffer may die before the
;; native call returns.  Every native call below therefore runs inside (keeping [everything whose address it passes, directly or inside
;; a struct] ...): the vector is held on the stack and fenced in a finally.  (ADVICE.md round 5: check-batch, set-full-indices, analysis.)
(defmacro ^:private keeping
  "Evaluates body with every object in the vector `xs` strongly reachable until body has returned."
  [xs & body]
  `(let [holder# ~xs]
     (try ~@body
          (finally (java.lang.ref.Reference/reachabilityFence holder#)))))

(defn- events-struct ^Memory [{:keys [n type process f a b]}]
  (doto (struct :events)
    (.setInt (o :events :n) n) (.setPointer (o :events :type) type) (.setPointer (o :events :process) process)
    (.setPointer (o :events :f) f) (.setPointer (o :events :a) a) (.setPointer (o :events :b) b)))

(defn- paired
  "tbc_pair_events: rows -> one op per invocation that may have taken effect (knossos.history/complete +
   without-failures + pairing).  Returns the op columns (Memory each) or nil on error."
  [{:keys [n pool] :as cols}]
  (let [cap   (max 1 n)
        of    (Memory. cap) oa (Memory. (* 4 cap)) ob (Memory. (* 4 cap)) op* (Memory. (* 4 cap))
        inv   (Memory. (* 4 cap)) ret (Memory. (* 4 cap))
        n-ops (IntByReference.) n-proc (IntByReference.)
        es    (events-struct cols)
        st    (keeping [cols es of oa ob op* inv ret n-ops n-proc]
                (.invokeInt (f "tbc_pair_events") (to-array [es of oa ob op* inv ret n-ops n-proc])))]
    (when (zero? st)
      (assoc cols :of of :oa oa :ob ob :oproc op* :inv inv :ret ret
                  :n-ops (.getValue n-ops) :n-process (.getValue n-proc)
                  :pool-mem (when pool (int-pool pool)) :pool-len (count pool)))))

(defn- ops-struct ^Memory [{:keys [n n-ops n-process of oa ob oproc inv ret pool-mem pool-len]}]
  (doto (struct :ops)
    (.setInt (o :ops :n) n-ops) (.setInt (o :ops :n_events) n)
    (.setPointer (o :ops :f) of) (.setPointer (o :ops :a) oa) (.setPointer (o :ops :b) ob)
    (.setPointer (o :ops :process) oproc) (.setPointer (o :ops :inv_pos) inv) (.setPointer (o :ops :ret_pos) ret)
    (.setPointer (o :ops :pool) (or pool-mem Pointer/NULL)) (.setInt (o :ops :pool_len) (int pool-len))
    (.setInt (o :ops :n_process) n-process)))

(defn- model-struct ^Memory [kind init n-keys]
  (doto (struct :model)
    (.setInt (o :model :kind) kind) (.setInt (o :model :init) init) (.setInt (o :model :n_keys) n-keys)))

(defn- opts-struct
  "All zero = the library's defaults.  :algorithm nil / :competition -> the level sweep for one history, handing over to
   the depth-first search when it must; :wgl -> the sequential knossos.wgl order; :linear -> the sweep."
  ^Memory [{:keys [time-limit device algorithm] :or {time-limit 0 device 0}}]
  (doto (struct :opts)
    (.setInt (o :opts :algorithm) (get-in abi [:enums :alg (or algorithm :competition)] 0))
    (.setInt (o :opts :device) device)
    (.setLong (o :opts :time_limit_ms) time-limit)))

;; ---------------------------------------------------------------------------------------------------------
;; tbc_result -> the Knossos result map
;; ---------------------------------------------------------------------------------------------------------
(defn- result-map
  "res: Memory at one tbc_result; base: its byte offset; ops / inv / ret: the history rows and the op -> row columns."
  [^Memory res ^long base m ops ^Memory inv ^Memory ret]
  (let [R          (fn [k] (+ base (o :result k)))
        valid      (.getInt res (R :valid))
        fail-op    (.getInt res (R :fail_op))
        prev-op    (.getInt res (R :prev_ok_op))
        completion (fn [i] (nth ops (.getInt ret (* 4 i))))          ; op index -> completion row -> the original op map
        invocation (fn [i] (nth ops (.getInt inv (* 4 i))))
        analyzer   (if (= (get-in abi [:enums :alg :linear]) (.getInt res (R :analyzer))) :linear :wgl)
        configs    (vec (for [c (range (min (get-in abi [:enums :max-final-configs]) (.getInt res (R :n_configs))))
                              :let [co (+ (R :configs) (* (o :config :size) c))
                                    st (.getInt res (+ co (o :config :state)))
                                    np (min (get-in abi [:enums :config-pending]) (.getInt res (+ co (o :config :n_pending))))
                                    lm (.getInt res (+ co (o :config :linearized_mask)))
                                    pend (fn [k] (invocation (.getInt res (+ co (o :config :pending) (* 4 k)))))]]
                          {:model      (state->model m st)
                           :last-op    (let [l (.getInt res (+ co (o :config :last_op)))] (when (not= l NO-OP) (invocation l)))
                           :pending    (vec (for [k (range np)] (pend k)))
                           :linearized (set (for [k (range np) :when (bit-test lm k)] (pend k)))}))]
    (case valid
      1  {:valid? true :analyzer analyzer
          :configs [{:model (state->model m (.getInt res (R :final_state))) :pending []}]
          :final-paths []}
      0  (let [x (completion fail-op)]
           {:valid? false :analyzer analyzer
            :op x
            :previous-ok (when (not= prev-op NO-OP) (completion prev-op))
            :configs (mapv #(dissoc % :linearized) configs)
            ;; knossos.linear's :final-paths: how each stuck config's last steps end in an inconsistent
            ;; model -- the failing op applied directly, or after one more pending call the model accepts
            ;; (same construction as knossos/_analysis.py::final_paths; model/step gives Knossos's own :msg)
            :final-paths
            (vec (take 10
                       (for [{:keys [model last-op pending linearized]} configs
                             path (cons [x]
                                        (for [y pending :when (and (not (linearized y)) (not= y x))] [y x]))
                             :let [steps  (reductions (fn [mm y] (model/step mm y)) model path)
                                   models (rest steps)]
                             :when (and (model/inconsistent? (last models))
                                        (not-any? model/inconsistent? (butlast models)))]
                         (vec (concat (when last-op [{:op last-op :model model}])
                                      (map (fn [y mm] {:op y :model mm}) path models))))))})
      {:valid? :unknown :analyzer analyzer
       :cause (get {1 :time-limit 2 :step-limit 3 :memory} (.getInt res (R :cause)) :unknown)})))

;; ---------------------------------------------------------------------------------------------------------
;; entry points
;; ---------------------------------------------------------------------------------------------------------
(defn analysis
  "(knossos.wgl/analysis model history) / (knossos.linear/analysis ...) / (knossos.competition/analysis ...) on the GPU.
   Returns nil when the GPU path is unavailable so that the caller can fall back."
  [m hist opts]
  (when-let [[kind init n-keys] (and @lib (model->native m))]
    (when-let [{:keys [ops inv ret] :as p} (paired (columns hist m))]
      (let [res  (doto (Memory. (o :result :size)) (.clear))
            os   (ops-struct p)                                    ; holds the ADDRESSES of p's columns, not the columns
            ms   (model-struct kind init n-keys)
            opt  (opts-struct opts)]
        (keeping [p os ms opt res]
          (let [st (.invokeInt (f "tbc_check") (to-array [os ms opt res]))]
            (try
              (when (zero? st)
                (result-map res 0 m ops inv ret))
              (finally (.invoke (f "tbc_result_free") Void/TYPE (to-array [res]))))))))))

(defn linearizable
  "Drop-in for (checker/linearizable {:model m :algorithm a}): same options, same result map
   ({:valid? :op :previous-ok :configs :final-paths :analyzer}).  Extra, additive keys: :device, :time-limit (ms).
   With crashed (:info) calls and :algorithm nil the library searches in the count form (DESIGN.md 2.4): :valid?, :op
   and :previous-ok are exact, :configs of an invalid verdict reached through the relaxed refutation holds the ONE config the
   prefix's linearization ended in (not every config stuck at the failing completion)."
  [{:keys [model] :as opts}]
  (let [stock (checker/linearizable opts)]
    (reify checker/Checker
      (check [_ test hist check-opts]
        (or (try (analysis model hist opts)
                 (catch Throwable t (warn t "libtbcheck failed; falling back to Knossos") nil))
            (checker/check stock test hist check-opts))))))

(defn check-batch
  "All keys of an independent/checker in ONE launch (tbc_batch_create / run / destroy): histories = one client-op
   vector per key.  Returns one result map per key, or nil.  Register-family models only (one tbc_model serves the
   batch; jepsen-tigerbeetle_amd/jepsen/independent.py::_check_batched is the tested version)."
  [m histories opts]
  (when-let [[kind init n-keys] (and @lib (model->native m))]
    (let [enc   (mapv #(paired (columns % m)) histories)]
      (when (every? some? enc)
        (let [offs   (vec (reductions + 0 (map :n-ops enc)))
              total  (max 1 (last offs))
              cat    (fn [k width]
                       (let [mem (Memory. (* width total))]
                         (doseq [[e off] (map vector enc offs) :let [len (* width (:n-ops e))] :when (pos? len)]
                           (.write mem (long (* width off)) (.getByteArray ^Memory (k e) 0 len) 0 len))
                         mem))
              nh     (count histories)
              op-off (let [mem (Memory. (* 8 (inc nh)))] (dorun (map-indexed (fn [i x] (.setLong mem (* 8 i) x)) offs)) mem)
              n-ev   (int-pool (map :n enc))
              n-pr   (int-pool (map :n-process enc))
              C      (o :batch_desc :cols)
              c-f    (cat :of 1) c-a (cat :oa 4) c-b (cat :ob 4) c-p (cat :oproc 4) c-inv (cat :inv 4) c-ret (cat :ret 4)
              desc   (doto (struct :batch_desc)
                       (.setInt (o :batch_desc :n_hist) nh) (.setPointer (o :batch_desc :op_off) op-off)
                       (.setPointer (o :batch_desc :n_events) n-ev) (.setPointer (o :batch_desc :n_process) n-pr)
                       (.setInt (+ C (o :ops :n)) (last offs))
                       (.setPointer (+ C (o :ops :f)) c-f) (.setPointer (+ C (o :ops :a)) c-a)
                       (.setPointer (+ C (o :ops :b)) c-b) (.setPointer (+ C (o :ops :process)) c-p)
                       (.setPointer (+ C (o :ops :inv_pos)) c-inv) (.setPointer (+ C (o :ops :ret_pos)) c-ret))
              ms     (model-struct kind init n-keys)
              opt    (opts-struct opts)
              hnd    (PointerByReference.)
              res    (doto (Memory. (* (o :result :size) (max 1 nh))) (.clear))]
          ;; (the library copies the columns during tbc_batch_create; the results and the per-key columns result-map reads live to the end)
          (keeping [enc op-off n-ev n-pr c-f c-a c-b c-p c-inv c-ret desc ms opt hnd res]
            (when (zero? (.invokeInt (f "tbc_batch_create") (to-array [desc ms opt hnd])))
              (let [keep? (::keep-open opts)          ; open-stream: the batch stays for check-next!
                    out   (try
                            (when (zero? (.invokeInt (f "tbc_batch_run") (to-array [(.getValue hnd) res])))
                              (vec (map-indexed (fn [i {:keys [ops inv ret]}] (result-map res (* i (o :result :size)) m ops inv ret)) enc)))
                            (finally (when-not keep? (.invoke (f "tbc_batch_destroy") Void/TYPE (to-array [(.getValue hnd)])))))]
                (if keep?
                  (if out
                    {:handle (.getValue hnd) :model m :n-hist-cap nh :results out}
                    (do (.invoke (f "tbc_batch_destroy") Void/TYPE (to-array [(.getValue hnd)])) nil))
                  out)))))))))

;; ---------------------------------------------------------------------------------------------------------
;; streaming: ONE batch kept across many inputs (include/tbcheck.h "streaming"; csrc/batch_stream.hip).  The reference checks every
;; history once (core.clj:139-146; workloads/set_full.clj:155-158): a run that checks many groups of keys opens a stream with the first
;; group and hands every later group to the same batch -- its arenas, streams and kernel choices stay, the op columns cross PCIe in
;; the 12-byte wire format, straight out of the slot this code writes them into.
;; ---------------------------------------------------------------------------------------------------------
(defn- wire-word
  "TBC_WIRE_WORD: f | a << 4 | b << 12 | process << 20 (a / b: 0..254 or nil = 255)."
  ^long [^long f a b ^long process]
  (let [nil8 (long (get-in abi [:enums :wire-nil]))
        v8   (fn [v] (cond (= v NIL) nil8 (<= 0 v 254) (long v) :else (throw (ex-info "value beyond the wire format" {:value v}))))]
    (when-not (<= 0 process 4095) (throw (ex-info "process beyond the wire format" {:process process})))
    (bit-or f (bit-shift-left (v8 a) 4) (bit-shift-left (if (= f (f-code :cas)) (v8 b) 0) 12) (bit-shift-left process 20))))

(defn open-stream
  "A batch created from the first group of histories (as check-batch) that stays open: returns {:handle :model :n-hist-cap :results}
   or nil.  Check further groups with check-next!, free it with close-stream!."
  [m histories opts]
  (when-let [first-results (check-batch m histories (assoc opts ::keep-open true))]
    first-results))

(defn check-next!
  "The next group of histories through an open stream (tbc_batch_map_input -> the wire columns written in place ->
   tbc_batch_submit_input -> tbc_batch_run).  `slot` 0 / 1 alternately lets the copy of one group run under the search of the
   previous one when two threads feed the stream.  Returns one result map per history, or nil (the caller then creates a batch)."
  [{:keys [handle model]} histories slot]
  (let [enc (mapv #(paired (columns % model)) histories)]
    (when (every? some? enc)
      (let [in  (struct :batch_input)
            nh  (count enc)
            res (doto (Memory. (* (o :result :size) (max 1 nh))) (.clear))]
        (keeping [enc in res]
          (when (and (zero? (.invokeInt (f "tbc_batch_map_input") (to-array [handle (int slot) in])))
                     (<= nh (.getInt in (o :batch_input :n_hist_cap)))
                     (<= (reduce + (map :n-ops enc)) (.getLong in (o :batch_input :ops_cap))))
            (let [op-off (.getPointer in (o :batch_input :op_off)) n-ev (.getPointer in (o :batch_input :n_events))
                  n-pr   (.getPointer in (o :batch_input :n_process)) word (.getPointer in (o :batch_input :word))
                  inv    (.getPointer in (o :batch_input :inv_pos))  ret  (.getPointer in (o :batch_input :ret_pos))]
              (.setLong op-off 0 0)
              (loop [es enc i 0 off 0]
                (when-let [{:keys [n n-ops n-process of oa ob oproc] :as e} (first es)]
                  (dotimes [j n-ops]
                    (let [at (* 4 (+ off j))]
                      (.setInt word at (unchecked-int (wire-word (.getByte ^Memory of j) (.getInt ^Memory oa (* 4 j)) (.getInt ^Memory ob (* 4 j))
                                                                 (.getInt ^Memory oproc (* 4 j)))))
                      (.setInt inv at (.getInt ^Memory (:inv e) (* 4 j)))
                      (.setInt ret at (.getInt ^Memory (:ret e) (* 4 j)))))
                  (.setInt n-ev (* 4 i) (int n)) (.setInt n-pr (* 4 i) (int n-process))
                  (.setLong op-off (* 8 (inc i)) (+ off n-ops))
                  (recur (rest es) (inc i) (+ off n-ops))))
              (when (and (zero? (.invokeInt (f "tbc_batch_submit_input") (to-array [handle (int slot) (int nh)])))
                         (zero? (.invokeInt (f "tbc_batch_run") (to-array [handle res]))))
                (vec (map-indexed (fn [i {:keys [ops inv ret]}] (result-map res (* i (o :result :size)) model ops inv ret)) enc))))))))))

(defn progress
  "How far the run that is out on an open stream has come -- the one call another thread may make on a handle while check-next! is in
   flight (tbc_batch_progress; knossos.search's reporter logs the same kind of line while a search runs):
   {:histories n :decided n :running? bool :elapsed-ms t}, or nil."
  [{:keys [handle]}]
  (let [p (struct :progress)]
    (keeping [p]
      (when (zero? (.invokeInt (f "tbc_batch_progress") (to-array [handle p])))
        {:histories  (.getInt p (o :progress :n_histories))
         :decided    (.getInt p (o :progress :n_decided))
         :running?   (not (zero? (.getInt p (o :progress :running))))
         :elapsed-ms (/ (.getLong p (o :progress :elapsed_ns)) 1e6)}))))

(defn close-stream! [{:keys [handle]}]
  (when handle (.invoke (f "tbc_batch_destroy") Void/TYPE (to-array [handle]))))

;; ---------------------------------------------------------------------------------------------------------
;; one history over several GPUs, one JVM per GPU (include/tbcheck.h tbc_comm_*; csrc/tbc_comm.hip): rank 0 makes the id, the
;; ranks exchange its 128 bytes however they talk to each other (the control node's own channel), every rank calls sharded-analysis
;; with the SAME history and gets the same result map.
;; ---------------------------------------------------------------------------------------------------------
(defn comm-unique-id
  "128 bytes for rank 0 to hand to the other ranks."
  ^bytes []
  (let [n (int (get-in abi [:enums :comm-id-bytes])) id (Memory. n)]
    (when (zero? (.invokeInt (f "tbc_comm_unique_id") (to-array [id])))
      (.getByteArray id 0 n))))

(defn comm-init
  "This rank's RCCL communicator (tbc_comm_init): a Pointer to keep and to free with comm-destroy!, or nil."
  [rank world ^bytes id device]
  (let [n (int (get-in abi [:enums :comm-id-bytes])) mem (doto (Memory. n) (.write 0 id 0 n)) out (PointerByReference.)]
    (keeping [mem out]
      (when (zero? (.invokeInt (f "tbc_comm_init") (to-array [(int rank) (int world) mem (int device) out])))
        (.getValue out)))))

(defn comm-destroy! [comm] (when comm (.invoke (f "tbc_comm_destroy") Void/TYPE (to-array [comm]))))

(defn sharded-analysis
  "(knossos.linear/analysis model history) with this rank's share of the level sweep's wavefronts and ONE all-gather of the
   relation tables inside the library (tbc_batch_sweep_allgather).  Every rank of `comm` calls it with the same history."
  [comm m hist opts]
  (when-let [[kind init n-keys] (and @lib comm (model->native m))]
    (when-let [{:keys [ops inv ret n-ops n n-process] :as p} (paired (columns hist m))]
      (let [os     (ops-struct p)
            op-off (let [mem (Memory. (* 8 2))] (dorun (map-indexed (fn [i x] (.setLong mem (* 8 i) x)) [0 n-ops])) mem)   ; op_off[0 .. 1] of the one history
            n-ev   (int-pool [n]) n-pr (int-pool [n-process])
            C      (o :batch_desc :cols)
            desc   (doto (struct :batch_desc)
                     (.setInt (o :batch_desc :n_hist) 1) (.setPointer (o :batch_desc :op_off) op-off)
                     (.setPointer (o :batch_desc :n_events) n-ev) (.setPointer (o :batch_desc :n_process) n-pr))
            _      (.write desc (long C) (.getByteArray os 0 (o :ops :size)) 0 (int (o :ops :size)))      ; tbc_batch_desc.cols = the tbc_ops
            ms     (model-struct kind init n-keys)
            opt    (opts-struct (assoc opts :algorithm :linear))
            hnd    (PointerByReference.)
            res    (doto (Memory. (o :result :size)) (.clear))]
        (keeping [p os op-off n-ev n-pr desc ms opt hnd res]
          (when (zero? (.invokeInt (f "tbc_batch_create") (to-array [desc ms opt hnd])))
            (try
              (when (zero? (.invokeInt (f "tbc_batch_sweep_allgather") (to-array [(.getValue hnd) comm res])))
                (result-map res 0 m ops inv ret))
              (finally (.invoke (f "tbc_batch_destroy") Void/TYPE (to-array [(.getValue hnd)]))))))))))

(defn memo-table
  "knossos.model.memo/memo through tbc_memo_build: the dense transition table of an arbitrary model over the op
   classes of this history (JNA callback = step) -> tbc_model kind :table."
  [m hist]
  (let [classes (vec (distinct (map #(select-keys % [:f :value]) (filter op/invoke? hist))))
        states  (atom [m])                                              ; handle -> model
        step    (reify Callback
                  (callback [_ ^long st ^int cls _user]
                    (let [m2 (model/step (@states st) (classes cls))]
                      (if (model/inconsistent? m2)
                        -1
                        (or (first (keep-indexed #(when (= %2 m2) %1) @states))
                            (dec (count (swap! states conj m2))))))))
        cap     65534
        table   (Memory. (* 2 cap (max 1 (count classes))))
        handles (Memory. (* 8 cap))
        n       (IntByReference.)]
    (when (zero? (.invokeInt (f "tbc_memo_build") (to-array [(long 0) (int (count classes)) step Pointer/NULL
                                                             (int cap) table handles n])))
      {:table table :n-states (.getValue n) :classes (zipmap classes (range))})))

;; ---------------------------------------------------------------------------------------------------------
;; jepsen.checker/set-full on the device (reference call site: workloads/set_full.clj:157)
;; ---------------------------------------------------------------------------------------------------------
(defn set-full-indices
  "history (one key, client ops only, :index = position) -> {:elements [...] :known [...] :last-present [...] :last-absent [...]}
   (op indices; 0xFFFFFFFF = none), computed by libtbcheck's scan (csrc/set_full.hip).  jepsen's own set-full-results
   arithmetic turns the three indices per element into the result map (jepsen/set_full.py::result_map is the tested statement)."
  [hist device]
  (let [adds     (->> hist (filter #(and (= :add (:f %)) (op/invoke? %))) (reduce (fn [mm x] (assoc mm (:value x) (:index x))) {}))
        elements (vec (map key (sort-by val adds)))                         ; numbered by their last :add invocation
        elem-no  (zipmap elements (range))
        add-ok   (reduce (fn [mm x] (if (and (= :add (:f x)) (op/ok? x) (>= (:index x) (get adds (:value x) Long/MAX_VALUE))
                                             (not (mm (:value x))))
                                      (assoc mm (:value x) (:index x)) mm)) {} hist)
        pairs    (history/pair-index hist)
        reads    (vec (for [x hist :when (and (op/invoke? x) (= :read (:f x)))
                            :let [c (pairs x)] :when (and c (op/ok? c) (some? (:value c)))]
                        [(:index x) (:index c) (:value c)]))                  ; :ok reads with a value, by invocation
        E (count elements) R (count reads) wpr (max 1 (quot (+ E 31) 32))
        present  (doto (Memory. (max 4 (* 4 R wpr))) (.clear))]
    (doseq [[r [_ _ v]] (map-indexed vector reads), x (distinct v) :let [e (elem-no x)] :when e]
      (let [off (* 4 (+ (* r wpr) (quot e 32)))]
        (.setInt present off (unchecked-int (bit-or (.getInt present off) (bit-shift-left 1 (rem e 32)))))))
    (let [a-inv (int-pool (map adds elements))
          a-ok  (int-pool (map #(get add-ok % NO-OP) elements))
          r-inv (int-pool (map first reads))
          r-ok  (int-pool (map second reads))
          in  (doto (struct :setfull_in)
                (.setInt (o :setfull_in :n_elements) E) (.setInt (o :setfull_in :n_reads) R)
                (.setInt (o :setfull_in :words_per_row) wpr) (.setInt (o :setfull_in :device) device)
                (.setPointer (o :setfull_in :add_invoke) a-inv)
                (.setPointer (o :setfull_in :add_ok) a-ok)
                (.setPointer (o :setfull_in :read_invoke) r-inv)
                (.setPointer (o :setfull_in :read_ok) r-ok)
                (.setPointer (o :setfull_in :present) present))
          hnd (PointerByReference.)
          k   (Memory. (max 4 (* 4 E))) lp (Memory. (max 4 (* 4 E))) la (Memory. (max 4 (* 4 E)))
          out (doto (struct :setfull_out)
                (.setPointer (o :setfull_out :known) k) (.setPointer (o :setfull_out :last_present) lp)
                (.setPointer (o :setfull_out :last_absent) la))]
      (keeping [a-inv a-ok r-inv r-ok present in hnd k lp la out]
        (when (zero? (.invokeInt (f "tbc_setfull_create") (to-array [in hnd])))       ; TBC_ERR_INVALID_ARG if the orders above are violated
          (try
            (when (zero? (.invokeInt (f "tbc_setfull_run") (to-array [(.getValue hnd) out])))
              {:elements elements :known (vec (.getIntArray k 0 E)) :last-present (vec (.getIntArray lp 0 E))
               :last-absent (vec (.getIntArray la 0 E))})
            (finally (.invoke (f "tbc_setfull_destroy") Void/TYPE (to-array [(.getValue hnd)])))))))))
