/*
 * bvgraph_hip_jni.c — JNI glue between integration/HipBVGraph.java and libbvgraph_hip.so (include/bvgraph_hip.h).  Complete source, every native written
 * out; NOT compiled in this repository (the image has no JDK: no jni.h).  Build on a box with one:
 *     cc -O2 -fPIC -shared -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -Iinclude -o libbvgraph_hip_jni.so integration/bvgraph_hip_jni.c \
 *        -Lwebgraph-big_amd/lib -lbvgraph_hip
 * Every function is one bvg_* call; no state lives here.  Statuses become the exceptions the reference throws at the same places:
 *     BVG_E_ARG          IllegalArgumentException          BVGraph.java:823,863,1000,1128   (node out of range)
 *     BVG_E_STATE        IllegalStateException             BVGraph.java:701 (reference > window), 832,1136 (no offsets)
 *     BVG_E_UNSUPPORTED  UnsupportedOperationException     BVGraph.java:631,658,699,733,763,794,864
 *     BVG_E_IO           IOException                       BVGraph.java:1492-1497,1326      (class / version / flag / file)
 *     BVG_E_EOF          EOFException                      the bit stream ends inside a record (dsiutils InputBitStream)
 *     BVG_E_NOMEM        OutOfMemoryError
 *     BVG_E_HIP          RuntimeException                  no gfx950 device / HIP failure: there is no CPU fallback
 * BVG_E_CAPACITY never becomes an exception: nDecodeRange / nSuccessorsBatch return -(needed) and the Java side grows its buffer.
 */
#include <jni.h>
#include <stdint.h>
#include <string.h>

#include "bvgraph_hip.h"

#define H(h) ((bvg_graph*)(intptr_t)(h))

static void throw_status(JNIEnv* e, int st) {
    const char* cls;
    switch (st) {
        case BVG_E_ARG: cls = "java/lang/IllegalArgumentException"; break;
        case BVG_E_STATE: cls = "java/lang/IllegalStateException"; break;
        case BVG_E_UNSUPPORTED: cls = "java/lang/UnsupportedOperationException"; break;
        case BVG_E_IO: cls = "java/io/IOException"; break;
        case BVG_E_EOF: cls = "java/io/EOFException"; break;
        case BVG_E_NOMEM: cls = "java/lang/OutOfMemoryError"; break;
        default: cls = "java/lang/RuntimeException"; break;
    }
    jclass c = (*e)->FindClass(e, cls);
    if (c) (*e)->ThrowNew(e, c, bvg_strerror(st));
}
static jlongArray longs(JNIEnv* e, const jlong* v, jsize n) {
    jlongArray a = (*e)->NewLongArray(e, n);
    if (a) (*e)->SetLongArrayRegion(e, a, 0, n, v);
    return a;
}

JNIEXPORT jlong JNICALL Java_it_unimi_dsi_big_webgraph_HipBVGraph_nOpen(JNIEnv* e, jclass c, jstring basename, jint mode, jint device) {
    (void)c;
    const char* s = (*e)->GetStringUTFChars(e, basename, 0);
    if (!s) return 0;
    bvg_graph* g = 0;
    const int st = bvg_open(s, mode, device, &g);
    (*e)->ReleaseStringUTFChars(e, basename, s);
    if (st) { throw_status(e, st); return 0; }
    return (jlong)(intptr_t)g;
}
JNIEXPORT jlong JNICALL Java_it_unimi_dsi_big_webgraph_HipBVGraph_nCopy(JNIEnv* e, jclass c, jlong h) {
    (void)c;
    bvg_graph* g = 0;
    const int st = bvg_copy(H(h), &g);
    if (st) { throw_status(e, st); return 0; }
    return (jlong)(intptr_t)g;
}
JNIEXPORT void JNICALL Java_it_unimi_dsi_big_webgraph_HipBVGraph_nClose(JNIEnv* e, jclass c, jlong h) { (void)e; (void)c; bvg_close(H(h)); }

JNIEXPORT jlongArray JNICALL Java_it_unimi_dsi_big_webgraph_HipBVGraph_nInfo(JNIEnv* e, jclass c, jlong h) {
    (void)c;
    bvg_params p;
    const int st = bvg_info(H(h), &p);
    if (st) { throw_status(e, st); return 0; }
    const jlong v[6] = { p.nodes, p.arcs, p.window_size, p.max_ref_count, p.min_interval_length, p.zeta_k };
    return longs(e, v, 6);
}
JNIEXPORT void JNICALL Java_it_unimi_dsi_big_webgraph_HipBVGraph_nOutdegrees(JNIEnv* e, jclass c, jlong h, jlong from, jlong to, jintArray out) {
    (void)c;
    if (to < from || (*e)->GetArrayLength(e, out) < to - from) { throw_status(e, BVG_E_ARG); return; }
    /* bvg_outdegrees launches kernels and waits for a stream: never inside a JNI critical region (it would hold the collector, and every Java thread that needs it,
       for the duration).  The degrees land in a page-locked block and are copied into the array afterwards. */
    const jsize k = (jsize)(to - from);
    int32_t* d = (int32_t*)bvg_host_alloc((size_t)(k > 0 ? k : 1) * sizeof(int32_t));
    if (!d) { throw_status(e, BVG_E_NOMEM); return; }
    const int st = bvg_outdegrees(H(h), from, to, d);
    if (!st) (*e)->SetIntArrayRegion(e, out, 0, k, (const jint*)d);
    bvg_host_free(d);
    if (st) throw_status(e, st);
}
/* direct (page-locked) buffers in, count out; -(needed) when the successor buffer is too small */
JNIEXPORT jlong JNICALL Java_it_unimi_dsi_big_webgraph_HipBVGraph_nDecodeRange(JNIEnv* e, jclass c, jlong h, jlong from, jlong to, jobject outdeg, jobject succ) {
    (void)c;
    int32_t* d = (int32_t*)(*e)->GetDirectBufferAddress(e, outdeg);
    int64_t* s = (int64_t*)(*e)->GetDirectBufferAddress(e, succ);
    if (!d || !s || (*e)->GetDirectBufferCapacity(e, outdeg) < to - from) { throw_status(e, BVG_E_ARG); return 0; }
    uint64_t n = 0;
    const int st = bvg_decode_range(H(h), from, to, d, s, (uint64_t)(*e)->GetDirectBufferCapacity(e, succ), &n);
    if (st == BVG_E_CAPACITY) return -(jlong)n;
    if (st) { throw_status(e, st); return 0; }
    return (jlong)n;
}
JNIEXPORT jlong JNICALL Java_it_unimi_dsi_big_webgraph_HipBVGraph_nSuccessorsBatch(JNIEnv* e, jclass c, jlong h, jlongArray nodes, jobject outdeg, jobject succ) {
    (void)c;
    const jsize k = (*e)->GetArrayLength(e, nodes);
    int32_t* d = (int32_t*)(*e)->GetDirectBufferAddress(e, outdeg);
    int64_t* s = (int64_t*)(*e)->GetDirectBufferAddress(e, succ);
    if (!d || !s || (*e)->GetDirectBufferCapacity(e, outdeg) < k) { throw_status(e, BVG_E_ARG); return 0; }
    jlong* x = (*e)->GetLongArrayElements(e, nodes, 0);
    if (!x) return 0;
    uint64_t n = 0;
    const int st = bvg_successors_batch(H(h), (const int64_t*)x, k, d, s, (uint64_t)(*e)->GetDirectBufferCapacity(e, succ), &n);
    (*e)->ReleaseLongArrayElements(e, nodes, x, JNI_ABORT);
    if (st == BVG_E_CAPACITY) return -(jlong)n;
    if (st) { throw_status(e, st); return 0; }
    return (jlong)n;
}
JNIEXPORT jobject JNICALL Java_it_unimi_dsi_big_webgraph_HipBVGraph_nHostAlloc(JNIEnv* e, jclass c, jlong bytes) {
    (void)c;
    void* p = bvg_host_alloc((size_t)(bytes > 0 ? bytes : 8));
    if (!p) { throw_status(e, BVG_E_NOMEM); return 0; }
    jobject b = (*e)->NewDirectByteBuffer(e, p, bytes > 0 ? bytes : 8);
    if (!b) bvg_host_free(p);                                  /* (an OutOfMemoryError is pending: the pinned block must not leak) */
    return b;
}
JNIEXPORT void JNICALL Java_it_unimi_dsi_big_webgraph_HipBVGraph_nHostFree(JNIEnv* e, jclass c, jobject b) {
    (void)c;
    if (b) bvg_host_free((*e)->GetDirectBufferAddress(e, b));
}
JNIEXPORT jlongArray JNICALL Java_it_unimi_dsi_big_webgraph_HipBVGraph_nScan(JNIEnv* e, jclass c, jlong h, jlong from, jlong to) {
    (void)c;
    bvg_scan_result r;
    const int st = bvg_scan(H(h), from, to, &r);
    if (st) { throw_status(e, st); return 0; }
    const jlong v[6] = { (jlong)r.nodes, (jlong)r.arcs, (jlong)r.chk, (jlong)r.graph_bytes, (jlong)r.index_bytes, (jlong)(r.kernel_ms * 1e6) };
    return longs(e, v, 6);
}
JNIEXPORT jlongArray JNICALL Java_it_unimi_dsi_big_webgraph_HipBVGraph_nScanMulti(JNIEnv* e, jclass c, jlongArray handles, jint balance) {
    (void)c;
    const jsize k = (*e)->GetArrayLength(e, handles);
    if (k < 1 || k > 64) { throw_status(e, BVG_E_ARG); return 0; }
    bvg_graph* g[64];
    jlong* hs = (*e)->GetLongArrayElements(e, handles, 0);
    if (!hs) return 0;
    for (jsize i = 0; i < k; i++) g[i] = H(hs[i]);
    (*e)->ReleaseLongArrayElements(e, handles, hs, JNI_ABORT);
    bvg_scan_result r;
    const int st = bvg_scan_multi(g, k, balance, &r, 0);
    if (st) { throw_status(e, st); return 0; }
    const jlong v[4] = { (jlong)r.nodes, (jlong)r.arcs, (jlong)r.chk, (jlong)(r.kernel_ms * 1e6) };
    return longs(e, v, 4);
}
JNIEXPORT jlongArray JNICALL Java_it_unimi_dsi_big_webgraph_HipBVGraph_nShardBounds(JNIEnv* e, jclass c, jlong h, jint k, jint balance) {
    (void)c;
    if (k < 1 || k > (1 << 20)) { throw_status(e, BVG_E_ARG); return 0; }
    jlongArray a = (*e)->NewLongArray(e, k + 1);
    if (!a) return 0;
    jlong* b = (*e)->GetLongArrayElements(e, a, 0);
    if (!b) return 0;
    const int st = bvg_shard_bounds(H(h), k, balance, (int64_t*)b);
    (*e)->ReleaseLongArrayElements(e, a, b, 0);
    if (st) { throw_status(e, st); return 0; }
    return a;
}
JNIEXPORT jlongArray JNICALL Java_it_unimi_dsi_big_webgraph_HipBVGraph_nBuildIndex(JNIEnv* e, jclass c, jlong h, jlong from, jlong to) {
    (void)c;
    uint64_t entries = 0, bytes = 0;
    const int st = bvg_build_index(H(h), from, to, &entries, &bytes);
    if (st) { throw_status(e, st); return 0; }
    const jlong v[2] = { (jlong)entries, (jlong)bytes };
    return longs(e, v, 2);
}
static void path_call(JNIEnv* e, jlong h, jstring path, int (*f)(bvg_graph*, const char*)) {
    const char* s = (*e)->GetStringUTFChars(e, path, 0);
    if (!s) return;
    const int st = f(H(h), s);
    (*e)->ReleaseStringUTFChars(e, path, s);
    if (st) throw_status(e, st);
}
JNIEXPORT void JNICALL Java_it_unimi_dsi_big_webgraph_HipBVGraph_nSaveIndex(JNIEnv* e, jclass c, jlong h, jstring path) { (void)c; path_call(e, h, path, bvg_save_index); }
JNIEXPORT void JNICALL Java_it_unimi_dsi_big_webgraph_HipBVGraph_nLoadIndex(JNIEnv* e, jclass c, jlong h, jstring path) { (void)c; path_call(e, h, path, bvg_load_index); }
/* mode: 0 = the full residual skip index, 1 = none (neither built nor read), 2 = marks only (validation marks + entries for lists of >= 4 096 residuals: bvg_tuning.no_index) */
JNIEXPORT void JNICALL Java_it_unimi_dsi_big_webgraph_HipBVGraph_nSetIndexMode(JNIEnv* e, jclass c, jlong h, jint mode) {
    (void)c;
    if (mode < 0 || mode > 2) { throw_status(e, BVG_E_ARG); return; }
    bvg_tuning t;
    memset(&t, 0, sizeof t);
    t.no_index = (uint32_t)mode;
    const int st = bvg_set_tuning(H(h), &t);
    if (st) throw_status(e, st);
}
