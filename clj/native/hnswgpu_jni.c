/* JNI alternative to the Panama binding in clj/src/hnsw/gpu.clj: a mechanical wrapper over the C ABI
 * (include/hnswgpu.h) for JVMs older than 22.  UNVERIFIED: there is no jni.h in this image, so this
 * file is not part of the build (compile with: gcc -shared -fPIC -I$JAVA_HOME/include
 * -I$JAVA_HOME/include/linux -I../../include hnswgpu_jni.c -L../../hnsw-clj_amd -lhnswgpu -o libhnswgpu_jni.so).
 * Java side: package hnsw.gpu; class Native { static native long create(float[] base, long n, int dim,
 * int metric, int device); static native int hnswBuild(long h, int M, int efc, long seed); static native
 * int hnswSearch(long h, float[] q, int nq, int k, int ef, int[] ids, float[] dist); ... } */
#include <jni.h>

#include "hnswgpu.h"

static void throw_last(JNIEnv *env) {
    jclass ex = (*env)->FindClass(env, "java/lang/RuntimeException");
    (*env)->ThrowNew(env, ex, hnswgpu_last_error());
}
/* a null array argument: IllegalArgumentException instead of a crashed JVM */
static jint throw_iae(JNIEnv *env, const char *msg) {
    jclass ex = (*env)->FindClass(env, "java/lang/IllegalArgumentException");
    if (ex) (*env)->ThrowNew(env, ex, msg);
    return HNSWGPU_EINVAL;
}

JNIEXPORT jlong JNICALL Java_hnsw_gpu_Native_create(JNIEnv *env, jclass c, jfloatArray base, jlong n, jint dim,
                                                    jint metric, jint device) {
    hnswgpu_index *idx = NULL;
    jfloat *p = (*env)->GetPrimitiveArrayCritical(env, base, NULL);
    int rc = hnswgpu_create(p, n, dim, metric, device, &idx); /* copies to HBM before returning */
    (*env)->ReleasePrimitiveArrayCritical(env, base, p, JNI_ABORT);
    if (rc != 0) throw_last(env);
    return (jlong)(intptr_t)idx;
}

JNIEXPORT void JNICALL Java_hnsw_gpu_Native_destroy(JNIEnv *env, jclass c, jlong h) {
    hnswgpu_destroy((hnswgpu_index *)(intptr_t)h);
}

JNIEXPORT jint JNICALL Java_hnsw_gpu_Native_hnswBuild(JNIEnv *env, jclass c, jlong h, jint M, jint efc, jlong seed) {
    int rc = hnswgpu_hnsw_build((hnswgpu_index *)(intptr_t)h, M, efc, seed);
    if (rc != 0) throw_last(env);
    return rc;
}

/* build options: include/hnswgpu.h HNSWGPU_BUILD_* (1 sequential, 2 heuristic, 4 symmetric, 8 extend) */
JNIEXPORT jint JNICALL Java_hnsw_gpu_Native_hnswBuildEx(JNIEnv *env, jclass c, jlong h, jint M, jint efc, jlong seed, jint flags) {
    int rc = hnswgpu_hnsw_build_ex((hnswgpu_index *)(intptr_t)h, M, efc, seed, flags);
    if (rc != 0) throw_last(env);
    return rc;
}

JNIEXPORT jint JNICALL Java_hnsw_gpu_Native_hnswSearch(JNIEnv *env, jclass c, jlong h, jfloatArray q, jint nq, jint k,
                                                       jint ef, jintArray ids, jfloatArray dist) {
    jfloat *pq = (*env)->GetFloatArrayElements(env, q, NULL);
    jint *pi = (*env)->GetIntArrayElements(env, ids, NULL);
    jfloat *pd = (*env)->GetFloatArrayElements(env, dist, NULL);
    int rc = hnswgpu_hnsw_search((hnswgpu_index *)(intptr_t)h, pq, nq, k, ef, (int32_t *)pi, pd, NULL);
    (*env)->ReleaseFloatArrayElements(env, q, pq, JNI_ABORT);
    (*env)->ReleaseIntArrayElements(env, ids, pi, 0);
    (*env)->ReleaseFloatArrayElements(env, dist, pd, 0);
    if (rc != 0) throw_last(env);
    return rc;
}

JNIEXPORT jint JNICALL Java_hnsw_gpu_Native_ivfBuild(JNIEnv *env, jclass c, jlong h, jint nlist, jint iters, jlong seed) {
    int rc = hnswgpu_ivf_build((hnswgpu_index *)(intptr_t)h, nlist, iters, seed);
    if (rc != 0) throw_last(env);
    return rc;
}

JNIEXPORT jint JNICALL Java_hnsw_gpu_Native_ivfSearch(JNIEnv *env, jclass c, jlong h, jfloatArray q, jint nq, jint k,
                                                      jint nprobe, jintArray ids, jfloatArray dist) {
    jfloat *pq = (*env)->GetFloatArrayElements(env, q, NULL);
    jint *pi = (*env)->GetIntArrayElements(env, ids, NULL);
    jfloat *pd = (*env)->GetFloatArrayElements(env, dist, NULL);
    int rc = hnswgpu_ivf_search((hnswgpu_index *)(intptr_t)h, pq, nq, k, nprobe, (int32_t *)pi, pd, NULL);
    (*env)->ReleaseFloatArrayElements(env, q, pq, JNI_ABORT);
    (*env)->ReleaseIntArrayElements(env, ids, pi, 0);
    (*env)->ReleaseFloatArrayElements(env, dist, pd, 0);
    if (rc != 0) throw_last(env);
    return rc;
}

/* ---- INTEGRATION.md section 5 / the simd-optimized seams / persistence: the same mechanical shape ---- */

JNIEXPORT jint JNICALL Java_hnsw_gpu_Native_setGraph(JNIEnv *env, jclass c, jlong h, jintArray levels, jintArray l0, jint M0,
                                                     jlongArray upOff, jintArray upAdj, jint M, jint entry, jint maxLevel) {
    if (!levels || !l0 || !upOff || !upAdj) return throw_iae(env, "setGraph: null array");
    jint *pl = (*env)->GetIntArrayElements(env, levels, NULL), *p0 = (*env)->GetIntArrayElements(env, l0, NULL);
    jlong *po = (*env)->GetLongArrayElements(env, upOff, NULL);
    jint *pu = (*env)->GetIntArrayElements(env, upAdj, NULL);
    if (!pl || !p0 || !po || !pu) { /* OutOfMemoryError is pending: release what was pinned and return */
        if (pl) (*env)->ReleaseIntArrayElements(env, levels, pl, JNI_ABORT);
        if (p0) (*env)->ReleaseIntArrayElements(env, l0, p0, JNI_ABORT);
        if (po) (*env)->ReleaseLongArrayElements(env, upOff, po, JNI_ABORT);
        if (pu) (*env)->ReleaseIntArrayElements(env, upAdj, pu, JNI_ABORT);
        return HNSWGPU_ENOMEM;
    }
    int rc = hnswgpu_set_graph((hnswgpu_index *)(intptr_t)h, (const int32_t *)pl, (const int32_t *)p0, M0, (const int64_t *)po,
                               (const int32_t *)pu, M, entry, maxLevel); /* validates and copies before returning */
    (*env)->ReleaseIntArrayElements(env, levels, pl, JNI_ABORT);
    (*env)->ReleaseIntArrayElements(env, l0, p0, JNI_ABORT);
    (*env)->ReleaseLongArrayElements(env, upOff, po, JNI_ABORT);
    (*env)->ReleaseIntArrayElements(env, upAdj, pu, JNI_ABORT);
    if (rc != 0) throw_last(env);
    return rc;
}

JNIEXPORT jint JNICALL Java_hnsw_gpu_Native_setIvf(JNIEnv *env, jclass c, jlong h, jfloatArray centroids, jint nlist,
                                                   jlongArray listOff, jintArray listIds) {
    if (!centroids || !listOff || !listIds) return throw_iae(env, "setIvf: null array");
    jfloat *pc = (*env)->GetFloatArrayElements(env, centroids, NULL);
    jlong *po = (*env)->GetLongArrayElements(env, listOff, NULL);
    jint *pi = (*env)->GetIntArrayElements(env, listIds, NULL);
    if (!pc || !po || !pi) {
        if (pc) (*env)->ReleaseFloatArrayElements(env, centroids, pc, JNI_ABORT);
        if (po) (*env)->ReleaseLongArrayElements(env, listOff, po, JNI_ABORT);
        if (pi) (*env)->ReleaseIntArrayElements(env, listIds, pi, JNI_ABORT);
        return HNSWGPU_ENOMEM;
    }
    int rc = hnswgpu_set_ivf((hnswgpu_index *)(intptr_t)h, pc, nlist, (const int64_t *)po, (const int32_t *)pi);
    (*env)->ReleaseFloatArrayElements(env, centroids, pc, JNI_ABORT);
    (*env)->ReleaseLongArrayElements(env, listOff, po, JNI_ABORT);
    (*env)->ReleaseIntArrayElements(env, listIds, pi, JNI_ABORT);
    if (rc != 0) throw_last(env);
    return rc;
}

JNIEXPORT jint JNICALL Java_hnsw_gpu_Native_batchDistances(JNIEnv *env, jclass c, jlong h, jfloatArray q, jintArray ids,
                                                           jint m, jfloatArray out) {
    if (!q || !out) return throw_iae(env, "batchDistances: null array");
    jfloat *pq = (*env)->GetFloatArrayElements(env, q, NULL);
    jint *pi = ids ? (*env)->GetIntArrayElements(env, ids, NULL) : NULL;
    jfloat *po = (*env)->GetFloatArrayElements(env, out, NULL);
    if (!pq || !po || (ids && !pi)) {
        if (pq) (*env)->ReleaseFloatArrayElements(env, q, pq, JNI_ABORT);
        if (pi) (*env)->ReleaseIntArrayElements(env, ids, pi, JNI_ABORT);
        if (po) (*env)->ReleaseFloatArrayElements(env, out, po, JNI_ABORT);
        return HNSWGPU_ENOMEM;
    }
    int rc = hnswgpu_batch_distances((hnswgpu_index *)(intptr_t)h, pq, (const int32_t *)pi, m, po);
    (*env)->ReleaseFloatArrayElements(env, q, pq, JNI_ABORT);
    if (pi) (*env)->ReleaseIntArrayElements(env, ids, pi, JNI_ABORT);
    (*env)->ReleaseFloatArrayElements(env, out, po, 0);
    if (rc != 0) throw_last(env);
    return rc;
}

JNIEXPORT jint JNICALL Java_hnsw_gpu_Native_save(JNIEnv *env, jclass c, jlong h, jstring path) {
    if (!path) return throw_iae(env, "save: null path");
    const char *p = (*env)->GetStringUTFChars(env, path, NULL);
    if (!p) return HNSWGPU_ENOMEM; /* OutOfMemoryError is pending */
    int rc = hnswgpu_save((hnswgpu_index *)(intptr_t)h, p);
    (*env)->ReleaseStringUTFChars(env, path, p);
    if (rc != 0) throw_last(env);
    return rc;
}

JNIEXPORT jlong JNICALL Java_hnsw_gpu_Native_load(JNIEnv *env, jclass c, jstring path, jint device) {
    hnswgpu_index *idx = NULL;
    if (!path) {
        throw_iae(env, "load: null path");
        return 0;
    }
    const char *p = (*env)->GetStringUTFChars(env, path, NULL);
    if (!p) return 0; /* OutOfMemoryError is pending */
    int rc = hnswgpu_load(p, device, &idx);
    (*env)->ReleaseStringUTFChars(env, path, p);
    if (rc != 0) throw_last(env);
    return (jlong)(intptr_t)idx;
}
