// mfsgd_jni.cpp -- JNI shim between java/MatrixFactorizationSGD.java and the C-ABI
// of include/mfsgd.h.
//
// UNCOMPILED AND UNTESTED: there is no JDK (no jni.h) in this container or on the
// GPU box.  Build, where a JDK exists:
//   g++ -std=c++17 -fPIC -shared -I$JAVA_HOME/include -I$JAVA_HOME/include/linux ...
//       -I../../include mfsgd_jni.cpp -L../lib -lmfsgd -Wl,-rpath,'$ORIGIN' -o libmfsgd_jni.so
//
// Rules followed: a Java array is pinned with
// Get/ReleasePrimitiveArrayCritical only around a plain memory copy the library
// makes on the host (set_factors); every call that can launch a
// kernel, start threads or wait for the device (set_ratings, train, predict,
// recommend) works on NATIVE copies made with Get*ArrayRegion, with no Java array
// pinned; array lengths are checked here, not only in Java; a failed native
// allocation throws OutOfMemoryError; a non-zero status becomes a RuntimeException
// carrying mfsgd_last_error(); no C++ exception crosses the boundary.
#include <jni.h>

#include <cstdint>
#include <memory>
#include <new>

#include "mfsgd.h"

namespace {

void throw_status(JNIEnv* env, mfsgd_handle* h, int rc) {
    if (rc == MFSGD_OK) return;
    jclass cls = env->FindClass("java/lang/RuntimeException");
    if (cls) env->ThrowNew(cls, mfsgd_last_error(h));
}

template <class T>
struct Pinned {
    JNIEnv* env;
    jarray arr;
    T* p;
    jint mode;
    Pinned(JNIEnv* e, jarray a, jint release_mode)
        : env(e), arr(a), p(a ? static_cast<T*>(e->GetPrimitiveArrayCritical(a, nullptr)) : nullptr), mode(release_mode) {}
    ~Pinned() {
        if (p) env->ReleasePrimitiveArrayCritical(arr, p, mode);
    }
};

mfsgd_handle* H(jlong h) { return reinterpret_cast<mfsgd_handle*>(h); }

void throw_new(JNIEnv* env, const char* cls_name, const char* msg) {
    jclass cls = env->FindClass(cls_name);
    if (cls) env->ThrowNew(cls, msg);
}

// new[] that reports failure to Java instead of throwing across the boundary
template <class T>
std::unique_ptr<T[]> alloc(JNIEnv* env, size_t n) {
    std::unique_ptr<T[]> p(new (std::nothrow) T[n ? n : 1]);
    if (!p) throw_new(env, "java/lang/OutOfMemoryError", "mfsgd_jni: native buffer");
    return p;
}

}  // namespace

extern "C" {

JNIEXPORT jlong JNICALL Java_MatrixFactorizationSGD_nativeCreate(JNIEnv* env, jclass, jint users, jint items, jint k,
                                                                 jfloat lr, jfloat lambda, jint device) {
    mfsgd_config cfg = {};
    cfg.n_users = users;
    cfg.n_items = items;
    cfg.k = k;
    cfg.lr = lr;
    cfg.lambda = lambda;
    cfg.device = device;
    mfsgd_handle* h = nullptr;
    throw_status(env, nullptr, mfsgd_create(&cfg, &h));
    return reinterpret_cast<jlong>(h);
}

JNIEXPORT void JNICALL Java_MatrixFactorizationSGD_nativeDestroy(JNIEnv*, jclass, jlong h) { mfsgd_destroy(H(h)); }

JNIEXPORT void JNICALL Java_MatrixFactorizationSGD_nativeSetRatings(JNIEnv* env, jclass, jlong h, jintArray u,
                                                                    jintArray i, jfloatArray r) {
    if (!u || !i || !r) return throw_new(env, "java/lang/NullPointerException", "ratings");
    const jsize n = env->GetArrayLength(u);
    if (env->GetArrayLength(i) != n || env->GetArrayLength(r) != n)
        return throw_new(env, "java/lang/IllegalArgumentException", "u, i and r must have the same length");
    // mfsgd_set_ratings hashes the triples, and when they are new it launches the ingest kernels, sorts on
    // the device and runs the packer on host threads (0.1 - 3 s): it gets native copies, nothing stays pinned
    auto cu = alloc<int32_t>(env, (size_t)n);
    auto ci = alloc<int32_t>(env, (size_t)n);
    auto cr = alloc<float>(env, (size_t)n);
    if (!cu || !ci || !cr) return;
    env->GetIntArrayRegion(u, 0, n, reinterpret_cast<jint*>(cu.get()));
    env->GetIntArrayRegion(i, 0, n, reinterpret_cast<jint*>(ci.get()));
    env->GetFloatArrayRegion(r, 0, n, cr.get());
    if (env->ExceptionCheck()) return;
    throw_status(env, H(h), mfsgd_set_ratings(H(h), cu.get(), ci.get(), cr.get(), n));
}

JNIEXPORT void JNICALL Java_MatrixFactorizationSGD_nativeInitFactors(JNIEnv* env, jclass, jlong h, jlong seed) {
    throw_status(env, H(h), mfsgd_init_factors(H(h), seed));
}

JNIEXPORT void JNICALL Java_MatrixFactorizationSGD_nativeSetFactors(JNIEnv* env, jclass, jlong h, jfloatArray p,
                                                                    jfloatArray q) {
    int32_t users = 0, items = 0, k = 0;
    if (!p || !q || mfsgd_get_dims(H(h), &users, &items, &k) != MFSGD_OK)
        return throw_new(env, "java/lang/IllegalArgumentException", "setFactors: bad handle or null array");
    if ((jlong)env->GetArrayLength(p) != (jlong)users * k || (jlong)env->GetArrayLength(q) != (jlong)items * k)
        return throw_new(env, "java/lang/IllegalArgumentException", "setFactors: P must be users x k, Q items x k");
    int rc;
    {
        // mfsgd_set_factors only copies into host staging (it may first wait for the device to go idle before
        // it frees the old device buffers -- no kernel is launched, nothing is allocated on the Java heap)
        Pinned<float> pp(env, p, JNI_ABORT), pq(env, q, JNI_ABORT);
        if (!pp.p || !pq.p) return throw_new(env, "java/lang/OutOfMemoryError", "mfsgd_jni: pin");
        rc = mfsgd_set_factors(H(h), pp.p, pq.p);
    }
    throw_status(env, H(h), rc);
}

JNIEXPORT void JNICALL Java_MatrixFactorizationSGD_nativeGetFactors(JNIEnv* env, jclass, jlong h, jfloatArray p,
                                                                    jfloatArray q) {
    int32_t users = 0, items = 0, k = 0;
    if (!p || !q || mfsgd_get_dims(H(h), &users, &items, &k) != MFSGD_OK)
        return throw_new(env, "java/lang/IllegalArgumentException", "factors: bad handle or null array");
    const size_t np = (size_t)users * k, nq = (size_t)items * k;
    if ((size_t)env->GetArrayLength(p) != np || (size_t)env->GetArrayLength(q) != nq)
        return throw_new(env, "java/lang/IllegalArgumentException", "factors: P must be users x k, Q items x k");
    // mfsgd_get_factors waits for the stream and copies from the device: native buffers, no pins
    auto cp = alloc<float>(env, np);
    auto cq = alloc<float>(env, nq);
    if (!cp || !cq) return;
    const int rc = mfsgd_get_factors(H(h), cp.get(), cq.get());
    if (rc == MFSGD_OK) {
        env->SetFloatArrayRegion(p, 0, (jsize)np, cp.get());
        env->SetFloatArrayRegion(q, 0, (jsize)nq, cq.get());
    }
    throw_status(env, H(h), rc);
}

JNIEXPORT void JNICALL Java_MatrixFactorizationSGD_nativeTrain(JNIEnv* env, jclass, jlong h, jint epochs,
                                                               jdoubleArray rmse) {
    if (epochs < 0 || !rmse || env->GetArrayLength(rmse) < epochs)
        return throw_new(env, "java/lang/IllegalArgumentException", "rmse array shorter than epochs");
    // the kernels run with NO Java array pinned: results land in a native buffer first
    auto tmp = alloc<double>(env, (size_t)epochs);
    if (!tmp) return;  // OutOfMemoryError pending: never train with the RMSE silently dropped
    const int rc = mfsgd_train(H(h), epochs, tmp.get());
    if (rc == MFSGD_OK && epochs > 0) env->SetDoubleArrayRegion(rmse, 0, epochs, tmp.get());
    throw_status(env, H(h), rc);
}

JNIEXPORT jdouble JNICALL Java_MatrixFactorizationSGD_nativeRmse(JNIEnv* env, jclass, jlong h) {
    double out = 0.0;
    throw_status(env, H(h), mfsgd_rmse(H(h), &out));
    return out;
}

JNIEXPORT void JNICALL Java_MatrixFactorizationSGD_nativePredict(JNIEnv* env, jclass, jlong h, jintArray u, jintArray i,
                                                                 jfloatArray out) {
    if (!u || !i || !out) return throw_new(env, "java/lang/NullPointerException", "predict");
    const jsize n = env->GetArrayLength(u);
    if (env->GetArrayLength(i) != n || env->GetArrayLength(out) < n)
        return throw_new(env, "java/lang/IllegalArgumentException", "u, i and out must have the same length");
    // copies, not pins: mfsgd_predict launches a kernel and waits for it
    auto cu = alloc<int32_t>(env, (size_t)n);
    auto ci = alloc<int32_t>(env, (size_t)n);
    auto co = alloc<float>(env, (size_t)n);
    if (!cu || !ci || !co) return;
    env->GetIntArrayRegion(u, 0, n, reinterpret_cast<jint*>(cu.get()));
    env->GetIntArrayRegion(i, 0, n, reinterpret_cast<jint*>(ci.get()));
    if (env->ExceptionCheck()) return;
    const int rc = mfsgd_predict(H(h), cu.get(), ci.get(), co.get(), n);
    if (rc == MFSGD_OK && n > 0) env->SetFloatArrayRegion(out, 0, n, co.get());
    throw_status(env, H(h), rc);
}

JNIEXPORT void JNICALL Java_MatrixFactorizationSGD_nativeRecommend(JNIEnv* env, jclass, jlong h, jintArray users,
                                                                   jint topn, jintArray items, jfloatArray scores) {
    if (!users || !items || !scores) return throw_new(env, "java/lang/NullPointerException", "recommend");
    const jsize n = env->GetArrayLength(users);
    if (topn < 1 || (jlong)env->GetArrayLength(items) < (jlong)n * topn || (jlong)env->GetArrayLength(scores) < (jlong)n * topn)
        return throw_new(env, "java/lang/IllegalArgumentException", "items / scores shorter than users x topN");
    auto cu = alloc<int32_t>(env, (size_t)n);
    auto ci = alloc<int32_t>(env, (size_t)n * (size_t)topn);
    auto cs = alloc<float>(env, (size_t)n * (size_t)topn);
    if (!cu || !ci || !cs) return;
    env->GetIntArrayRegion(users, 0, n, reinterpret_cast<jint*>(cu.get()));
    if (env->ExceptionCheck()) return;
    const int rc = mfsgd_recommend(H(h), cu.get(), n, topn, ci.get(), cs.get());
    if (rc == MFSGD_OK && n > 0) {
        env->SetIntArrayRegion(items, 0, n * topn, reinterpret_cast<const jint*>(ci.get()));
        env->SetFloatArrayRegion(scores, 0, n * topn, cs.get());
    }
    throw_status(env, H(h), rc);
}

}  // extern "C"
