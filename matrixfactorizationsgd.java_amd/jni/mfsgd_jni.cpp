// mfsgd_jni.cpp -- JNI shim between java/MatrixFactorizationSGD.java and the C-ABI
// of include/mfsgd.h.
//
// UNCOMPILED AND UNTESTED: there is no JDK (no jni.h) in this container or on the
// GPU box.  Build, where a JDK exists:
//   g++ -std=c++17 -fPIC -shared -I$JAVA_HOME/include -I$JAVA_HOME/include/linux \
//       -I../../include mfsgd_jni.cpp -L../lib -lmfsgd -Wl,-rpath,'$ORIGIN' -o libmfsgd_jni.so
//
// Rules followed: arrays are pinned with
// Get/ReleasePrimitiveArrayCritical only around the library's copy-in/copy-out,
// never across a kernel launch; a non-zero status becomes a RuntimeException
// carrying mfsgd_last_error(); no C++ exception crosses the boundary.
#include <jni.h>

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
    const jsize n = env->GetArrayLength(u);
    int rc;
    {
        // host-only work (bucketing + step packing) on the pinned arrays; JNI_ABORT: read-only
        Pinned<int32_t> pu(env, u, JNI_ABORT), pi(env, i, JNI_ABORT);
        Pinned<float> pr(env, r, JNI_ABORT);
        rc = mfsgd_set_ratings(H(h), pu.p, pi.p, pr.p, n);
    }
    throw_status(env, H(h), rc);
}

JNIEXPORT void JNICALL Java_MatrixFactorizationSGD_nativeInitFactors(JNIEnv* env, jclass, jlong h, jlong seed) {
    throw_status(env, H(h), mfsgd_init_factors(H(h), seed));
}

JNIEXPORT void JNICALL Java_MatrixFactorizationSGD_nativeSetFactors(JNIEnv* env, jclass, jlong h, jfloatArray p,
                                                                    jfloatArray q) {
    int rc;
    {
        Pinned<float> pp(env, p, JNI_ABORT), pq(env, q, JNI_ABORT);
        rc = mfsgd_set_factors(H(h), pp.p, pq.p);
    }
    throw_status(env, H(h), rc);
}

JNIEXPORT void JNICALL Java_MatrixFactorizationSGD_nativeGetFactors(JNIEnv* env, jclass, jlong h, jfloatArray p,
                                                                    jfloatArray q) {
    int rc;
    {
        Pinned<float> pp(env, p, 0), pq(env, q, 0);
        rc = mfsgd_get_factors(H(h), pp.p, pq.p);
    }
    throw_status(env, H(h), rc);
}

JNIEXPORT void JNICALL Java_MatrixFactorizationSGD_nativeTrain(JNIEnv* env, jclass, jlong h, jint epochs,
                                                               jdoubleArray rmse) {
    // the kernels run with NO Java array pinned: results land in a native buffer first
    double* tmp = epochs > 0 ? new (std::nothrow) double[epochs] : nullptr;
    const int rc = mfsgd_train(H(h), epochs, tmp);
    if (rc == MFSGD_OK && tmp) env->SetDoubleArrayRegion(rmse, 0, epochs, tmp);
    delete[] tmp;
    throw_status(env, H(h), rc);
}

JNIEXPORT jdouble JNICALL Java_MatrixFactorizationSGD_nativeRmse(JNIEnv* env, jclass, jlong h) {
    double out = 0.0;
    throw_status(env, H(h), mfsgd_rmse(H(h), &out));
    return out;
}

JNIEXPORT void JNICALL Java_MatrixFactorizationSGD_nativePredict(JNIEnv* env, jclass, jlong h, jintArray u, jintArray i,
                                                                 jfloatArray out) {
    const jsize n = env->GetArrayLength(u);
    // copies, not pins: mfsgd_predict launches a kernel and waits for it
    jint* cu = env->GetIntArrayElements(u, nullptr);
    jint* ci = env->GetIntArrayElements(i, nullptr);
    float* co = n > 0 ? new (std::nothrow) float[n] : nullptr;
    const int rc = mfsgd_predict(H(h), reinterpret_cast<const int32_t*>(cu), reinterpret_cast<const int32_t*>(ci), co, n);
    env->ReleaseIntArrayElements(u, cu, JNI_ABORT);
    env->ReleaseIntArrayElements(i, ci, JNI_ABORT);
    if (rc == MFSGD_OK && co) env->SetFloatArrayRegion(out, 0, n, co);
    delete[] co;
    throw_status(env, H(h), rc);
}

JNIEXPORT void JNICALL Java_MatrixFactorizationSGD_nativeRecommend(JNIEnv* env, jclass, jlong h, jintArray users,
                                                                   jint topn, jintArray items, jfloatArray scores) {
    const jsize n = env->GetArrayLength(users);
    jint* cu = env->GetIntArrayElements(users, nullptr);
    int32_t* ci = n > 0 ? new (std::nothrow) int32_t[(size_t)n * topn] : nullptr;
    float* cs = n > 0 ? new (std::nothrow) float[(size_t)n * topn] : nullptr;
    const int rc = mfsgd_recommend(H(h), reinterpret_cast<const int32_t*>(cu), n, topn, ci, cs);
    env->ReleaseIntArrayElements(users, cu, JNI_ABORT);
    if (rc == MFSGD_OK && ci && cs) {
        env->SetIntArrayRegion(items, 0, n * topn, reinterpret_cast<const jint*>(ci));
        env->SetFloatArrayRegion(scores, 0, n * topn, cs);
    }
    delete[] ci;
    delete[] cs;
    throw_status(env, H(h), rc);
}

}  // extern "C"
