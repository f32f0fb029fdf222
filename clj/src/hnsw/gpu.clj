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
     (search-ivf index query-vec k & {:keys [num-probes]})"
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
(def ^:private h-build    (delay (fn-handle "hnswgpu_hnsw_build" (FunctionDescriptor/of I (into-array [P I I L])))))
(def ^:private h-search   (delay (fn-handle "hnswgpu_hnsw_search" (FunctionDescriptor/of I (into-array [P P I I I P P P])))))
(def ^:private h-ivfbuild (delay (fn-handle "hnswgpu_ivf_build" (FunctionDescriptor/of I (into-array [P I I L])))))
(def ^:private h-ivfsearch (delay (fn-handle "hnswgpu_ivf_search" (FunctionDescriptor/of I (into-array [P P I I I P P P])))))
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

(defn build-index
  "hnsw.ultra-fast/build-index (src/hnsw/ultra_fast.clj:334-344): data = seq of [id ^doubles vector]."
  [data & {:keys [M ef-construction metric seed] :or {M 16 ef-construction 200 metric :cosine seed 42}}]
  (let [idx (create data metric :hnsw)]
    (check (.invokeWithArguments ^MethodHandle @h-build [(:handle idx) (int M) (int ef-construction) (long seed)]))
    idx))

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

(defn close! [idx]
  (check (.invokeWithArguments ^MethodHandle @h-destroy [(:handle idx)])))
