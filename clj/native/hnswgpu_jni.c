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
