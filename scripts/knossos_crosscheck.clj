;; knossos_crosscheck.clj -- pins this repository's expectations against STOCK Knossos, from outside.
;;
;; Nothing in this repository's environment can run Clojure (no JVM; SURVEY.md section 0, F4), so parity with
;; stock Knossos is "unpinned" (DESIGN.md).  Anyone with a JVM can close that gap:
;;
;;   clojure -Sdeps '{:deps {knossos/knossos {:mvn/version "0.3.8"} cheshire/cheshire {:mvn/version "5.11.0"}}}' \
;;           -M scripts/knossos_crosscheck.clj tests/golden/edn > stock-knossos.json
;;   python scripts/compare_crosscheck.py stock-knossos.json        ; prints every disagreement
;;
;; and, for the checkers the reference actually runs (tests/golden/edn_checkers: jepsen.checker/set-full, knossos.model/set):
;;
;;   clojure -Sdeps '{:deps {jepsen/jepsen {:mvn/version "0.2.7"} cheshire/cheshire {:mvn/version "5.11.0"}}}' \
;;           -M scripts/knossos_crosscheck.clj tests/golden/edn_checkers > stock-checkers.json
;;   python scripts/compare_crosscheck.py --checkers stock-checkers.json
;;
;; For every history file it runs knossos.wgl/analysis and knossos.linear/analysis with the model named in
;; expected.json and prints one JSON object per file: {file, model, valid?, op-index, analyzer results}.
;; op-index = the :index of the :op Knossos reports for an invalid history (the completion that cannot be
;; linearized).  The history files hold one op map per line, exactly what jepsen.store writes to history.edn
;; (the reference keeps them under store/, /root/reference/.gitignore:7).
(require '[clojure.edn :as edn]
         '[clojure.java.io :as io]
         '[clojure.string :as str]
         '[cheshire.core :as json]
         '[knossos.model :as model]
         '[knossos.wgl :as wgl]
         '[knossos.linear :as linear])

(def models {"cas-register" #(model/cas-register)
             "register"     #(model/register)
             "mutex"        #(model/mutex)
             "set"          #(model/set)})          ; (no "bank": Knossos ships none -- those files pin this repository's own model)

;; jepsen.checker is only needed for the checkers directory: resolved at run time so that the plain Knossos run needs no jepsen
(defn set-full-result [history linearizable?]
  (let [check   (requiring-resolve 'jepsen.checker/check)
        setfull (requiring-resolve 'jepsen.checker/set-full)
        r       (check (setfull {:linearizable? linearizable?}) {} history {})]
    (select-keys r [:valid? :attempt-count :stable-count :lost-count :lost :never-read-count :never-read
                    :stale-count :stale :duplicated-count :duplicated :stable-latencies :lost-latencies])))

(defn read-history [file]
  (with-open [r (io/reader file)]
    (->> (line-seq r)
         (remove str/blank?)
         (map edn/read-string)
         (filter #(integer? (:process %)))      ; the reference drops :nemesis rows the same way (tests/ledger.clj:94)
         vec)))

(defn summarize [a]
  {:valid? (:valid? a)
   :op-index (some-> a :op :index)
   :previous-ok-index (some-> a :previous-ok :index)
   :configs (count (:configs a))
   :final-paths (count (:final-paths a))})

(let [dir      (or (first *command-line-args*) "tests/golden/edn")
      expected (json/parse-string (slurp (io/file dir "expected.json")) true)]
  (doseq [{:keys [file model checker opts] :as c} (:cases expected)]
    (let [h (read-history (io/file dir file))]
      (cond
        (= checker "set-full")
        (println (json/generate-string {:file file :checker "set-full" :opts opts :provenance "stock-jepsen"
                                        :result (set-full-result h (:linearizable? opts))}))
        (nil? (models model))
        (println (json/generate-string {:file file :model model :skipped "no such model in Knossos"}))
        :else
        (let [m ((models model))
              w (summarize (wgl/analysis m h))
              l (summarize (linear/analysis m h))]
          (println (json/generate-string {:file file :model model :provenance "stock-knossos"
                                          :valid? (:valid? l) :op-index (:op-index l)
                                          :wgl w :linear l})))))))
