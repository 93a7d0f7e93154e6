/* maelsim_jni.c — thin JNI shim over include/maelsim.h for the reference's host language (Clojure/JVM).
 * NOT compiled in this image (no jni.h / JDK); build on a box with a JDK:
 *   cc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -Iinclude integration/jni/maelsim_jni.c \
 *      -Lmaelstrom_amd -lmaelsim -o libmaelsim_jni.so
 * Java side: class maelstrom.gpu.Native { static native byte[] configDefaults(int workload, int nNodes);
 *                                         static native long create(byte[] cfg, int device); ... }
 * Buffers cross as direct ByteBuffers over the engine-owned pinned host memory (valid until the next run). */
#include <jni.h>
#include <stdlib.h>
#include <string.h>
#include "maelsim.h"

static void throw_msim(JNIEnv *env, const char *msg) {
  (*env)->ThrowNew(env, (*env)->FindClass(env, "java/lang/RuntimeException"), msg);
}

/* The whole msim_config crosses as bytes (little-endian, the layout of include/maelsim.h: struct_size bytes), so every field —
 * key_count, max_txn_length, proxy_service, consistency_model, journal_capacity, capacities — is reachable from the JVM side.
 * configDefaults(workload, n_nodes) -> byte[struct_size] with the CLI defaults of core.clj:136-229 filled in. */
JNIEXPORT jbyteArray JNICALL Java_maelstrom_gpu_Native_configDefaults(JNIEnv *env, jclass cls, jint workload, jint n_nodes) {
  msim_config c;
  jbyteArray out;
  (void)cls;
  if (msim_config_defaults(&c, (uint32_t)workload, (uint32_t)n_nodes) != MSIM_OK) { throw_msim(env, "msim_config_defaults"); return NULL; }
  out = (*env)->NewByteArray(env, (jsize)sizeof c);
  (*env)->SetByteArrayRegion(env, out, 0, (jsize)sizeof c, (const jbyte *)&c);
  return out;
}

/* create(cfg bytes, device): validates and finalizes the config (msim_config_finalize) and builds the engine context */
JNIEXPORT jlong JNICALL Java_maelstrom_gpu_Native_create(JNIEnv *env, jclass cls, jbyteArray cfg_bytes, jint device) {
  msim_config c;
  char err[256];
  msim_ctx *ctx = NULL;
  (void)cls;
  if ((*env)->GetArrayLength(env, cfg_bytes) != (jsize)sizeof c) { throw_msim(env, "msim_config: wrong size (struct_size mismatch)"); return 0; }
  (*env)->GetByteArrayRegion(env, cfg_bytes, 0, (jsize)sizeof c, (jbyte *)&c);
  if (msim_create(&c, device, &ctx, err, sizeof err) != MSIM_OK) { throw_msim(env, err); return 0; }
  return (jlong)(intptr_t)ctx;
}

JNIEXPORT void JNICALL Java_maelstrom_gpu_Native_run(JNIEnv *env, jclass cls, jlong h, jlong first, jint n) {
  msim_ctx *ctx = (msim_ctx *)(intptr_t)h;
  (void)cls;
  if (msim_run(ctx, (uint64_t)first, (uint32_t)n) || msim_check(ctx) || msim_fetch(ctx)) throw_msim(env, msim_last_error(ctx));
}

/* returns {rows ByteBuffer (16 B each), payload ByteBuffer (u32)} of instance i */
JNIEXPORT jobjectArray JNICALL Java_maelstrom_gpu_Native_history(JNIEnv *env, jclass cls, jlong h, jint i) {
  msim_ctx *ctx = (msim_ctx *)(intptr_t)h;
  const msim_op *ops; const uint32_t *pay; uint32_t n_ops, n_words;
  jobjectArray out;
  (void)cls;
  if (msim_history(ctx, (uint32_t)i, &ops, &n_ops, &pay, &n_words)) { throw_msim(env, msim_last_error(ctx)); return NULL; }
  out = (*env)->NewObjectArray(env, 2, (*env)->FindClass(env, "java/nio/ByteBuffer"), NULL);
  (*env)->SetObjectArrayElement(env, out, 0, (*env)->NewDirectByteBuffer(env, (void *)ops, (jlong)n_ops * 16));
  (*env)->SetObjectArrayElement(env, out, 1, (*env)->NewDirectByteBuffer(env, (void *)pay, (jlong)n_words * 4));
  return out;
}

/* the history of instance i as history.edn text (msim_history_edn_rows): (jepsen.history/parse ...) or a store file, no decoder needed */
JNIEXPORT jstring JNICALL Java_maelstrom_gpu_Native_historyEdn(JNIEnv *env, jclass cls, jlong h, jint i) {
  msim_ctx *ctx = (msim_ctx *)(intptr_t)h;
  const msim_op *ops; const uint32_t *pay; uint32_t n_ops, n_words;
  msim_config c; size_t need = 0; char *buf; jstring out;
  (void)cls;
  if (msim_history(ctx, (uint32_t)i, &ops, &n_ops, &pay, &n_words) || msim_get_config(ctx, &c)) { throw_msim(env, msim_last_error(ctx)); return NULL; }
  if (msim_history_edn_rows(&c, ops, n_ops, pay, n_words, NULL, 0, &need) || !(buf = (char *)malloc(need))) { throw_msim(env, "history edn"); return NULL; }
  msim_history_edn_rows(&c, ops, n_ops, pay, n_words, buf, need, NULL);
  out = (*env)->NewStringUTF(env, buf);   /* the text is ASCII */
  free(buf);
  return out;
}

/* 6 longs: all/clients/servers x send/recv (net/checker.clj:28-41) */
JNIEXPORT jlongArray JNICALL Java_maelstrom_gpu_Native_netStats(JNIEnv *env, jclass cls, jlong h, jint i) {
  msim_net_stats st; jlongArray out = (*env)->NewLongArray(env, 6);
  (void)cls;
  if (msim_net_stats_get((msim_ctx *)(intptr_t)h, (uint32_t)i, &st)) { throw_msim(env, "net stats"); return NULL; }
  (*env)->SetLongArrayRegion(env, out, 0, 6, (const jlong *)&st);
  return out;
}

JNIEXPORT void JNICALL Java_maelstrom_gpu_Native_destroy(JNIEnv *env, jclass cls, jlong h) {
  (void)env; (void)cls;
  msim_destroy((msim_ctx *)(intptr_t)h);
}
