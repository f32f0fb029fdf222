(ns hnsw.gpu
  "MI355X engine behind hnsw-clj's index/search API (libhnswgpu.so, C ABI: include/hnswgpu.h).

   Binding: java.lang.foreign (Panama FFM, JDK 22+; the reference already requires Java 21+),
   so no native glue is needed.  A JNI alternative is in clj/native/hnswgpu_jni.c.

   UNVERIFIED: this image has no JVM, so this namespace has never been loaded.  The C ABI it calls is
   exercised end to end by tests/test_gpu_parity.py through ctypes with the same signatures.

   Drop-in surface (same names / argument meaning as the reference):
     (build-index data & {:keys [M ef-construction distance-fn]})   ; hnsw.ultra-fast/build-index
     (search-knn index query-vec k)                                  ; hnsw.ultra-fast/search-knn
     (search-batch index queries k)                                  ; BatchSearchIndex/search-batch*
     (build-ivf-index data & {:keys [num-partitions max-iterations]}) ; hnsw.ann.partition.ivf-flat/build-index
     (search-ivf index query-vec k & {:keys [num-probes]})
     (search-ivf-batch index queries k & {:keys [num-probes]})
     (batch-distances index query-vec)                               ; simd-optimized/batch-cosine-distances
     (top-k-distances index query-vec k)                             ; simd-optimized/top-k-distances
     (add-vector! index new-data)                                    ; hnsw.api/add-vector!, insert-single on a live index
     (from-ultra-graph graph)                                        ; a graph built by the reference's own insert-single
     (from-ivf-flat-index ivf)                                       ; an IVFFlatIndex built by the reference's own k-means
     (save idx path) / (load-index path ids)                         ; helper/index-io save-index / load-index
   and, at the bottom, the extend-type that makes GpuIndex an ANNIndex / BatchSearchIndex / PersistableIndex next to the
   records src/hnsw/api/unified.clj:30-95 extends."
  (:require [hnsw.api.protocol :as proto])
  (:import [java.lang.foreign Arena FunctionDescriptor Linker MemorySegment SymbolLookup ValueLayout]
           [java.lang.invoke MethodHandle]))

(def ^:private ^Linker linker (Linker/nativeLinker))
(def ^:private lookup
  (delay (SymbolLookup/libraryLookup (or (System/getProperty "hnswgpu.lib") "libhnswgpu.so") (Arena/global))))

(defn- fn-handle ^MethodHandle [^String sym ^FunctionDescriptor desc]
  (.downcallHandle linker (.orElseThrow (.find ^SymbolLookup @lookup sym)) desc (make-array java.lang.foreign.Linker$Option 0)))

(def ^:private I ValueLayout/JAVA_INT)
(def ^:private L ValueLayout/JAVA_LONG)
(def ^:private P ValueLayout/ADDRESS)

(def ^:private h-create   (delay (fn-handle "hnswgpu_create" (FunctionDescriptor/of I (into-array [P L I I I P])))))
(def ^:private h-destroy  (delay (fn-handle "hnswgpu_destroy" (FunctionDescriptor/of I (into-array [P])))))
(def ^:private h-build    (delay (fn-handle "hnswgpu_hnsw_build_ex" (FunctionDescriptor/of I (into-array [P I I L I])))))
(def ^:private h-search   (delay (fn-handle "hnswgpu_hnsw_search" (FunctionDescriptor/of I (into-array [P P I I I P P P])))))
(def ^:private h-ivfbuild (delay (fn-handle "hnswgpu_ivf_build" (FunctionDescriptor/of I (into-array [P I I L])))))
(def ^:private h-ivfsearch (delay (fn-handle "hnswgpu_ivf_search" (FunctionDescriptor/of I (into-array [P P I I I P P P])))))
(def ^:private h-setgraph (delay (fn-handle "hnswgpu_set_graph" (FunctionDescriptor/of I (into-array [P P P I P P I I I])))))
(def ^:private h-setivf   (delay (fn-handle "hnswgpu_set_ivf" (FunctionDescriptor/of I (into-array [P P I P P])))))
(def ^:private h-batchd   (delay (fn-handle "hnswgpu_batch_distances" (FunctionDescriptor/of I (into-array [P P P I P])))))
(def ^:private h-exact    (delay (fn-handle "hnswgpu_exact_knn" (FunctionDescriptor/of I (into-array [P P I I P P])))))
(def ^:private h-save     (delay (fn-handle "hnswgpu_save" (FunctionDescriptor/of I (into-array [P P])))))
(def ^:private h-load     (delay (fn-handle "hnswgpu_load" (FunctionDescriptor/of I (into-array [P I P])))))
(def ^:private h-info     (delay (fn-handle "hnswgpu_info" (FunctionDescriptor/of I (into-array [P P P P P P])))))
(def ^:private h-rejmode  (delay (fn-handle "hnswgpu_set_rejection_test" (FunctionDescriptor/of I (into-array [P I])))))
(def ^:private h-error    (delay (fn-handle "hnswgpu_last_error" (FunctionDescriptor/of P (into-array ValueLayout [])))))

(defn- check [rc]
  (when-not (zero? (int rc))
    (let [^MemorySegment msg (.invokeWithArguments ^MethodHandle @h-error [])]
      (throw (ex-info (str "libhnswgpu: " (.getString (.reinterpret msg 512) 0)) {:code rc})))))

(def ^:private metric-of {:cosine 0 :l2 1 :euclidean 1 :dot 2})

(defrecord GpuIndex [handle ids dim kind])

(defn- floats-of ^MemorySegment [^Arena arena rows dim]
  ;; [id double-array] pairs -> one contiguous float32 matrix (the engine stores f32)
  (let [n (count rows)
        seg (.allocate arena (* 4 (long n) (long dim)) 16)]
    (dotimes [i n]
      (let [^doubles v (nth rows i)]
        (dotimes [j dim]
          (.setAtIndex seg ValueLayout/JAVA_FLOAT (+ (* (long i) dim) j) (float (aget v j))))))
    seg))

(defn- create [data metric kind]
  (with-open [arena (Arena/ofConfined)]
    (let [ids (mapv first data)
          vecs (mapv second data)
          dim (if (seq vecs) (alength ^doubles (first vecs)) 1)
          base (floats-of arena vecs dim)
          out (.allocate arena 8 8)]
      (check (.invokeWithArguments ^MethodHandle @h-create
                                   [base (long (count ids)) (int dim) (int (metric-of metric 0)) (int 0) out]))
      (->GpuIndex (.get out P 0) ids dim kind))))

(def ^:private build-flag {:sequential 1 :heuristic 2 :symmetric 4 :extend 8})   ; include/hnswgpu.h: HNSWGPU_BUILD_*

(defn build-index
  "hnsw.ultra-fast/build-index (src/hnsw/ultra_fast.clj:334-344): data = seq of [id ^doubles vector].
   :select :closest   the m closest candidates (insert-single / prune-connections-ultra, ultra_fast.clj:216-299; default)
           :heuristic get-neighbors-heuristic for a node's links and for an over-full list (src/hnsw/graph.clj:162-232),
                      the dropped edge leaving the pruned list only
           :graph-clj the same, the dropped edge removed from both lists (prune-connections, graph.clj:226-231): what
                      hnsw.ann.graph.pure-hnsw/build-index builds
   :sequential? true  insert-single's own order and start level (slow: one launch round trip per row)."
  [data & {:keys [M ef-construction metric seed select sequential?]
           :or {M 16 ef-construction 200 metric :cosine seed 42 select :closest sequential? false}}]
  (let [idx (create data metric :hnsw)
        flags (bit-or (case select :closest 0 :heuristic 2 :graph-clj 6)
                      (if sequential? (build-flag :sequential) 0))]
    (check (.invokeWithArguments ^MethodHandle @h-build [(:handle idx) (int M) (int ef-construction) (long seed) (int flags)]))
    idx))

(def ^:private h-add (delay (fn-handle "hnswgpu_hnsw_add" (FunctionDescriptor/of I (into-array [P P L I L])))))

(defn add-vector!
  "hnsw.api/add-vector! (src/hnsw/api.clj:30-33) / ultra-fast insert-single (src/hnsw/ultra_fast.clj:216-275) on a live
   GpuIndex: the new [id ^doubles vector] pairs join the base matrix and the graph; returns the index with their ids
   appended (a GpuIndex is a value: keep the returned one).  One call per batch of new vectors is the efficient shape."
  [idx new-data & {:keys [ef-construction seed] :or {ef-construction 200 seed 42}}]
  (with-open [arena (Arena/ofConfined)]
    (check (.invokeWithArguments ^MethodHandle @h-add
                                 [(:handle idx) (floats-of arena (mapv second new-data) (:dim idx)) (long (count new-data))
                                  (int ef-construction) (long seed)])))
  (update idx :ids into (map first new-data)))

(defn- results [idx ^MemorySegment ids ^MemorySegment ds q k]
  (vec (for [i (range k)
             :let [id (.getAtIndex ids I (+ (* (long q) k) i))]
             :when (>= id 0)]
         {:id (nth (:ids idx) id) :distance (double (.getAtIndex ds ValueLayout/JAVA_FLOAT (+ (* (long q) k) i)))})))

(defn search-batch
  "All queries in one kernel launch -> vector of result vectors (BatchSearchIndex/search-batch*,
   src/hnsw/api/protocol.clj:58-67; replaces helper/parallel_search.clj:15-49)."
  [idx queries k & {:keys [ef] :or {ef 0}}]
  (with-open [arena (Arena/ofConfined)]
    (let [nq (count queries)
          q (floats-of arena (vec queries) (:dim idx))
          ids (.allocate arena (* 4 nq k) 4)
          ds (.allocate arena (* 4 nq k) 4)]
      (check (.invokeWithArguments ^MethodHandle @h-search
                                   [(:handle idx) q (int nq) (int k) (int ef) ids ds MemorySegment/NULL]))
      (mapv #(results idx ids ds % k) (range nq)))))

(defn search-knn
  "hnsw.ultra-fast/search-knn (src/hnsw/ultra_fast.clj:346-374): seq of {:id :distance} ascending."
  [idx ^doubles query-vec k]
  (first (search-batch idx [query-vec] k)))

(defn search-batch-routed
  "search-batch with the crossover the device offers and the reference has no word for: the traversal evaluates E(ef) rows per
   query, gathered at random; the exact scan (hnswgpu_exact_knn: every row once per batch through the matrix cores) answers at
   recall 1.0 -- the cheaper way to at least the same recall once E(ef) >= n / 3 (31k x 768: 1.5M QPS exact against 0.78M through
   the graph at ef 640, 90k at ef 3200).  E(ef) comes from the traversal's own counters (the `stats` argument of
   hnswgpu_hnsw_search: evals, expansions per query) on up to 32 of the queries.  The neighbours of a routed batch are the
   exact ones: at least as good as the graph's, not necessarily the same (hnsw-clj_amd/ultra_fast.py: search_batch route=True)."
  [idx queries k & {:keys [ef] :or {ef 0}}]
  (with-open [arena (Arena/ofConfined)]
    (let [pilot (vec (take 32 queries))
          np (count pilot)
          q (floats-of arena pilot (:dim idx))
          ids (.allocate arena (* 4 np k) 4)
          ds (.allocate arena (* 4 np k) 4)
          st (.allocate arena (* 16 np) 8)]
      (check (.invokeWithArguments ^MethodHandle @h-search [(:handle idx) q (int np) (int k) (int ef) ids ds st]))
      (let [evals (/ (reduce + (map #(.getAtIndex st ValueLayout/JAVA_LONG (* 2 (long %))) (range np))) (double np))]
        (if (>= evals (/ (count (:ids idx)) 3.0))
          (let [nq (count queries)
                qa (floats-of arena (vec queries) (:dim idx))
                ia (.allocate arena (* 4 nq k) 4)
                da (.allocate arena (* 4 nq k) 4)]
            (check (.invokeWithArguments ^MethodHandle @h-exact [(:handle idx) qa (int nq) (int k) ia da]))
            (mapv #(results idx ia da % k) (range nq)))
          (search-batch idx queries k :ef ef))))))

(defn build-ivf-index
  "hnsw.ann.partition.ivf-flat/build-index (src/hnsw/ann/partition/ivf_flat.clj:137-211,300-303)."
  [data & {:keys [num-partitions max-iterations metric] :or {num-partitions 24 max-iterations 10 metric :cosine}}]
  (let [idx (create data metric :ivf)]
    (check (.invokeWithArguments ^MethodHandle @h-ivfbuild [(:handle idx) (int num-partitions) (int max-iterations) (long 42)]))
    idx))

(defn search-ivf
  "search-ivf-flat (ivf_flat.clj:236-294) with an explicit :num-probes (mode presets :243-247 map to 1/2/4/8/12)."
  [idx ^doubles query-vec k & {:keys [num-probes] :or {num-probes 4}}]
  (with-open [arena (Arena/ofConfined)]
    (let [q (floats-of arena [query-vec] (:dim idx))
          ids (.allocate arena (* 4 k) 4)
          ds (.allocate arena (* 4 k) 4)]
      (check (.invokeWithArguments ^MethodHandle @h-ivfsearch
                                   [(:handle idx) q (int 1) (int k) (int num-probes) ids ds MemorySegment/NULL]))
      (results idx ids ds 0 k))))

(defn search-ivf-batch
  "All queries against the inverted lists in one call (the batch seam of search-ivf-flat)."
  [idx queries k & {:keys [num-probes] :or {num-probes 4}}]
  (with-open [arena (Arena/ofConfined)]
    (let [nq (count queries)
          q (floats-of arena (vec queries) (:dim idx))
          ids (.allocate arena (* 4 nq k) 4)
          ds (.allocate arena (* 4 nq k) 4)]
      (check (.invokeWithArguments ^MethodHandle @h-ivfsearch
                                   [(:handle idx) q (int nq) (int k) (int num-probes) ids ds MemorySegment/NULL]))
      (mapv #(results idx ids ds % k) (range nq)))))

;; ===== the simd-optimized batch seams =====

(defn batch-distances
  "simd-optimized/batch-cosine-distances / batch-euclidean-distances [query vectors] (src/hnsw/simd_optimized.clj:164-184)
   against the vectors the index already holds: distances from query-vec to every row, in row order (a double-array)."
  ^doubles [idx ^doubles query-vec]
  (with-open [arena (Arena/ofConfined)]
    (let [n (count (:ids idx))
          q (floats-of arena [query-vec] (:dim idx))
          out (.allocate arena (* 4 (long n)) 4)]
      (check (.invokeWithArguments ^MethodHandle @h-batchd [(:handle idx) q MemorySegment/NULL (int n) out]))
      (let [res (double-array n)]
        (dotimes [i n] (aset res i (double (.getAtIndex out ValueLayout/JAVA_FLOAT (long i)))))
        res))))

(defn top-k-distances
  "simd-optimized/top-k-distances (src/hnsw/simd_optimized.clj:271-280): exact k nearest rows -> [[row-index distance] ...]."
  [idx ^doubles query-vec k]
  (with-open [arena (Arena/ofConfined)]
    (let [q (floats-of arena [query-vec] (:dim idx))
          ids (.allocate arena (* 4 k) 4)
          ds (.allocate arena (* 4 k) 4)]
      (check (.invokeWithArguments ^MethodHandle @h-exact [(:handle idx) q (int 1) (int k) ids ds]))
      (vec (for [i (range k)
                 :let [id (.getAtIndex ids I (long i))]
                 :when (>= id 0)]
             [id (double (.getAtIndex ds ValueLayout/JAVA_FLOAT (long i)))])))))

;; ===== INTEGRATION.md section 5: serve an index the reference itself built =====

(defn- ints-of ^MemorySegment [^Arena arena coll]
  (let [seg (.allocate arena (* 4 (long (max 1 (count coll)))) 4)]
    (doseq [[i v] (map-indexed vector coll)] (.setAtIndex seg I (long i) (int v)))
    seg))

(defn- longs-of ^MemorySegment [^Arena arena coll]
  (let [seg (.allocate arena (* 8 (long (max 1 (count coll)))) 8)]
    (doseq [[i v] (map-indexed vector coll)] (.setAtIndex seg L (long i) (long v)))
    seg))

(defn from-ultra-graph
  "An UltraGraph built by the reference's own insert-single (src/hnsw/ultra_fast.clj:216-275) -> GpuIndex.
   Flattens UltraNode{id vector level neighbors} (:99-102; neighbors = Object[level+1] of HashSet<String>) into the
   arrays of include/hnswgpu.h: ids -> rows in iteration order of the node map; every neighbour set padded with -1
   to M0 = max-M (level 0) / M (levels >= 1); up_off = prefix sum of the node levels."
  [graph & {:keys [metric] :or {metric :cosine}}]
  (let [nodes (vec (.values ^java.util.Map (.nodes graph)))
        row-of (into {} (map-indexed (fn [i nd] [(.id nd) i]) nodes))
        M (int (.M graph))
        M0 (int (.max-M graph))
        idx (create (mapv (fn [nd] [(.id nd) (.vector nd)]) nodes) metric :hnsw)
        levels (mapv #(int (.level %)) nodes)
        up-off (vec (reductions + 0 levels))
        ;; a neighbour id the node map does not hold, or a set larger than the layer's width, is a broken graph: say so
        ;; here instead of an NPE inside ints-of / a silently truncated HashSet
        pad (fn [ids width]
              (let [rows (mapv (fn [id] (or (row-of id)
                                            (throw (ex-info "neighbour id is not a node of the graph" {:id id}))))
                               ids)]
                (when (> (count rows) width)
                  (throw (ex-info "neighbour set larger than the layer's width (M0 / M)" {:size (count rows) :width width})))
                (take width (concat rows (repeat -1)))))
        nbrs (fn [nd lv] (seq ^java.util.Set (aget ^objects (.neighbors nd) (int lv))))
        l0 (mapcat #(pad (nbrs % 0) M0) nodes)
        up (mapcat (fn [nd] (mapcat #(pad (nbrs nd %) M) (range 1 (inc (.level nd))))) nodes)
        entry (row-of (.get ^java.util.concurrent.atomic.AtomicReference (.entry-point graph)))]
    (with-open [arena (Arena/ofConfined)]
      (check (.invokeWithArguments ^MethodHandle @h-setgraph
                                   [(:handle idx) (ints-of arena levels) (ints-of arena l0) M0
                                    (longs-of arena up-off) (ints-of arena up) M (int entry) (int (nth levels entry))])))
    idx))

(defn from-ivf-flat-index
  "An IVFFlatIndex built by the reference's own k-means (src/hnsw/ann/partition/ivf_flat.clj:137-211) -> GpuIndex:
   centroids + list membership go to hnswgpu_set_ivf (rows = the partitions' vectors, partition by partition)."
  [ivf & {:keys [metric] :or {metric :cosine}}]
  (let [parts (:partitions ivf)
        rows (vec (mapcat identity parts))                 ; [[id ^doubles vec] ...], list by list
        idx (create rows metric :ivf)
        off (vec (reductions + 0 (map count parts)))]
    (with-open [arena (Arena/ofConfined)]
      (check (.invokeWithArguments ^MethodHandle @h-setivf
                                   [(:handle idx) (floats-of arena (vec (:centroids ivf)) (:dim idx)) (int (count parts))
                                    (longs-of arena off) (ints-of arena (range (count rows)))])))
    idx))

;; ===== persistence (helper/index-io save-index / load-index, src/hnsw/helper/index_io.clj:10-80) =====

(defn set-rejection-test!
  "Tuning knob with no counterpart in the reference: the engine decides neighbours / candidates that cannot reach the
   result from an int8 copy of the rows (+25 % memory) and reads f32 rows only for the rest; results never depend on it.
   mode :off (no copy), :auto (default: large HNSW batches, IVF batches of 9+ queries, dim >= 128) or :always."
  [idx mode]
  (check (.invokeWithArguments ^MethodHandle @h-rejmode [(:handle idx) (int ({:off 0 :auto 1 :always 2} mode))]))
  idx)

(defn save
  "One flat binary file (base + graph + lists); the String ids go beside it as EDN -- the engine knows rows only."
  [idx ^String path]
  (with-open [arena (Arena/ofConfined)]
    (check (.invokeWithArguments ^MethodHandle @h-save [(:handle idx) (.allocateFrom arena path)])))
  (spit (str path ".ids.edn") (pr-str {:ids (:ids idx) :kind (:kind idx)}))
  true)

(defn load-index
  "Index instance from `save`'s files; nil if the file does not exist (index_io.clj:41-48 returns nil too)."
  [^String path & {:keys [device] :or {device 0}}]
  (when (.exists (java.io.File. path))
    (with-open [arena (Arena/ofConfined)]
      (let [out (.allocate arena 8 8)
            dim (.allocate arena 4 4)]
        (check (.invokeWithArguments ^MethodHandle @h-load [(.allocateFrom arena path) (int device) out]))
        (let [h (.get out P 0)
              {:keys [ids kind]} (read-string (slurp (str path ".ids.edn")))]
          (check (.invokeWithArguments ^MethodHandle @h-info
                                       [h MemorySegment/NULL dim MemorySegment/NULL MemorySegment/NULL MemorySegment/NULL]))
          (->GpuIndex h ids (.get dim I 0) kind))))))

(defn close! [idx]
  (check (.invokeWithArguments ^MethodHandle @h-destroy [(:handle idx)])))

;; ===== ONE index over several GPUs (include/hnswgpu.h: hnswgpu_group_*) =====
;; search-partitioned's scatter / per-partition top-k / gather / sort / take k (src/hnsw/ann/partition/
;; partitioned_hnsw.clj:149-196) as one native call: the group owns one engine handle per GPU.

(def ^:private h-gcreate  (delay (fn-handle "hnswgpu_group_create" (FunctionDescriptor/of I (into-array [P I I I P])))))
(def ^:private h-gdestroy (delay (fn-handle "hnswgpu_group_destroy" (FunctionDescriptor/of I (into-array [P])))))
(def ^:private h-gsetivf  (delay (fn-handle "hnswgpu_group_set_ivf" (FunctionDescriptor/of I (into-array [P P L P I P P])))))
(def ^:private h-givf     (delay (fn-handle "hnswgpu_group_ivf_search" (FunctionDescriptor/of I (into-array [P P I I I P P])))))
(def ^:private h-gbuild   (delay (fn-handle "hnswgpu_group_hnsw_build" (FunctionDescriptor/of I (into-array [P P L I I L])))))
(def ^:private h-ghnsw    (delay (fn-handle "hnswgpu_group_hnsw_search" (FunctionDescriptor/of I (into-array [P P I I I P P])))))

(defrecord GpuGroup [handle ids dim kind devices])

(defn- group-create [devices dim metric]
  (with-open [arena (Arena/ofConfined)]
    (let [out (.allocate arena 8 8)]
      (check (.invokeWithArguments ^MethodHandle @h-gcreate
                                   [(ints-of arena devices) (int (count devices)) (int dim) (int (metric-of metric 0)) out]))
      (.get out P 0))))

(defn build-partitioned-index
  "hnsw.ann.partition.partitioned-hnsw/build-partitioned-hnsw (partitioned_hnsw.clj:46-143) across GPUs: contiguous row
   ranges, one HNSW sub-graph per device of `devices` (e.g. (range 8)), built on the devices."
  [data devices & {:keys [M ef-construction metric seed] :or {M 16 ef-construction 200 metric :cosine seed 42}}]
  (with-open [arena (Arena/ofConfined)]
    (let [ids (mapv first data)
          vecs (mapv second data)
          dim (alength ^doubles (first vecs))
          h (group-create (vec devices) dim metric)]
      (check (.invokeWithArguments ^MethodHandle @h-gbuild
                                   [h (floats-of arena vecs dim) (long (count ids)) (int M) (int ef-construction) (long seed)]))
      (->GpuGroup h ids dim :hnsw (vec devices)))))

(defn group-from-ivf-flat-index
  "An IVFFlatIndex (the reference's own k-means, ivf_flat.clj:137-211) served by several GPUs: centroids replicated, whole
   inverted lists dealt to the devices by row count.  Answers equal the one-GPU index's bit for bit."
  [ivf devices & {:keys [metric] :or {metric :cosine}}]
  (with-open [arena (Arena/ofConfined)]
    (let [parts (:partitions ivf)
          rows (vec (mapcat identity parts))
          dim (alength ^doubles (second (first rows)))
          off (vec (reductions + 0 (map count parts)))
          h (group-create (vec devices) dim metric)]
      (check (.invokeWithArguments ^MethodHandle @h-gsetivf
                                   [h (floats-of arena (mapv second rows) dim) (long (count rows))
                                    (floats-of arena (vec (:centroids ivf)) dim) (int (count parts))
                                    (longs-of arena off) (ints-of arena (range (count rows)))]))
      (->GpuGroup h (mapv first rows) dim :ivf (vec devices)))))

(defn search-group-batch
  "All queries in one call over all devices -> vector of result vectors.  :ivf groups take :num-probes, :hnsw groups :ef."
  [grp queries k & {:keys [num-probes ef] :or {num-probes 4 ef 50}}]
  (with-open [arena (Arena/ofConfined)]
    (let [nq (count queries)
          q (floats-of arena (vec queries) (:dim grp))
          ids (.allocate arena (* 4 nq k) 4)
          ds (.allocate arena (* 4 nq k) 4)
          ivf? (= :ivf (:kind grp))]
      (check (.invokeWithArguments ^MethodHandle (if ivf? @h-givf @h-ghnsw)
                                   [(:handle grp) q (int nq) (int k) (int (if ivf? num-probes (max ef k))) ids ds]))
      (mapv #(results grp ids ds % k) (range nq)))))

(defn close-group! [grp]
  (check (.invokeWithArguments ^MethodHandle @h-gdestroy [(:handle grp)])))

;; ===== first-class index next to the reference's own records (src/hnsw/api/unified.clj:30-95) =====

(def ^:private mode->probes {:turbo 1 :fast 2 :balanced 4 :accurate 8 :precise 12})   ; ivf_flat.clj:243-247

(extend-type GpuIndex
  proto/ANNIndex
  (search-knn* [this query k mode]
    (if (= :ivf (:kind this))
      (search-ivf this query k :num-probes (mode->probes mode 4))
      (search-knn this query k)))                          ; ef = (max k 50) whatever the mode, as graph.clj:304
  (index-info* [this] {:type (if (= :ivf (:kind this)) "GPU IVF-FLAT (MI355X)" "GPU HNSW (MI355X)")
                       :vectors (count (:ids this)) :dim (:dim this)})
  (index-type* [this] (if (= :ivf (:kind this)) :gpu-ivf-flat :gpu-hnsw))
  proto/BatchSearchIndex
  (search-batch* [this queries k mode]
    (if (= :ivf (:kind this))
      (search-ivf-batch this queries k :num-probes (mode->probes mode 4))
      (search-batch this queries k)))
  proto/PersistableIndex
  (save-index* [this filepath] (save this filepath)))
