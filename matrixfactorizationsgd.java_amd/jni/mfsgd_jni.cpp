// mfsgd_jni.cpp -- JNI shim between java/MatrixFactorizationSGD.java and the C-ABI
// of include/mfsgd.h.
//
// UNCOMPILED AND UNTESTED: there is no JDK (no jni.h) in this container or on the
// GPU box.  Build, where a JDK exists:
//   g++ -std=c++17 -fPIC -shared -I$JAVA_HOME/include -I$JAVA_HOME/include/linux ...
//       -I../../include mfsgd_jni.cpp -L../lib -lmfsgd -Wl,-rpath,'$ORIGIN' -o libmfsgd_jni.so
//
// Rules followed: NO Java array is ever pinned
// (Get/ReleasePrimitiveArrayCritical is not used: every C-ABI call may launch a
// kernel, start threads or wait for the device -- even set_factors waits for the
// device to go idle before it frees the old buffers -- and no JNI call may be made
// inside a critical region); every call works on NATIVE copies made with
// Get*ArrayRegion; array lengths are checked here, not only in Java; a failed native
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

mfsgd_handle* H(jlong h) { return reinterpret_cast<mfsgd_handle*>(h); }
mfsgd_dsgd* D(jlong d) { return reinterpret_cast<mfsgd_dsgd*>(d); }

void throw_dsgd(JNIEnv* env, mfsgd_dsgd* d, int rc) {
    if (rc == MFSGD_OK) return;
    jclass cls = env->FindClass("java/lang/RuntimeException");
    if (cls) env->ThrowNew(cls, mfsgd_dsgd_last_error(d));
}

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
                                                                 jfloat lr, jfloat lambda, jint device, jint n_parts) {
    mfsgd_config cfg = {};
    cfg.n_users = users;
    cfg.n_items = items;
    cfg.k = k;
    cfg.lr = lr;
    cfg.lambda = lambda;
    cfg.device = device;
    cfg.n_parts = n_parts;
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
    // native copies, nothing pinned: mfsgd_set_factors waits for the device to go idle before it frees the old
    // device buffers, and a pinned array would keep the collector out for as long as that takes
    const size_t np = (size_t)users * k, nq = (size_t)items * k;
    auto cp = alloc<float>(env, np);
    auto cq = alloc<float>(env, nq);
    if (!cp || !cq) return;
    env->GetFloatArrayRegion(p, 0, (jsize)np, cp.get());
    env->GetFloatArrayRegion(q, 0, (jsize)nq, cq.get());
    if (env->ExceptionCheck()) return;
    const int rc = mfsgd_set_factors(H(h), cp.get(), cq.get());
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

// ---- DSGD: the ring under the C-ABI (mfsgd_dsgd_*) for MatrixFactorizationSGD.trainDistributed ------------------

JNIEXPORT jbyteArray JNICALL Java_MatrixFactorizationSGD_nativeDsgdUniqueId(JNIEnv* env, jclass) {
    signed char id[MFSGD_DSGD_ID_BYTES];
    const int rc = mfsgd_dsgd_unique_id(id);
    if (rc != MFSGD_OK) {
        throw_dsgd(env, nullptr, rc);
        return nullptr;
    }
    jbyteArray out = env->NewByteArray(MFSGD_DSGD_ID_BYTES);
    if (out) env->SetByteArrayRegion(out, 0, MFSGD_DSGD_ID_BYTES, id);
    return out;  // null with OutOfMemoryError pending if the allocation failed
}

JNIEXPORT void JNICALL Java_MatrixFactorizationSGD_nativeDsgdPlan(JNIEnv* env, jclass, jlongArray deg_user, jlongArray deg_item,
                                                                  jint n_parts, jintArray user_begin, jintArray item_part) {
    if (!deg_user || !deg_item || !user_begin || !item_part) return throw_new(env, "java/lang/NullPointerException", "plan");
    const jsize nu = env->GetArrayLength(deg_user), ni = env->GetArrayLength(deg_item);
    if (n_parts < 1 || env->GetArrayLength(user_begin) != n_parts + 1 || env->GetArrayLength(item_part) != ni)
        return throw_new(env, "java/lang/IllegalArgumentException", "plan: userBegin needs nParts + 1 entries, itemPart one per item");
    auto du = alloc<int64_t>(env, (size_t)nu);
    auto di = alloc<int64_t>(env, (size_t)ni);
    auto ub = alloc<int32_t>(env, (size_t)n_parts + 1);
    auto ip = alloc<int32_t>(env, (size_t)ni);
    if (!du || !di || !ub || !ip) return;
    env->GetLongArrayRegion(deg_user, 0, nu, reinterpret_cast<jlong*>(du.get()));
    env->GetLongArrayRegion(deg_item, 0, ni, reinterpret_cast<jlong*>(di.get()));
    if (env->ExceptionCheck()) return;
    if (mfsgd_dsgd_plan(du.get(), di.get(), nu, ni, n_parts, ub.get(), ip.get()) != MFSGD_OK)
        return throw_new(env, "java/lang/IllegalArgumentException", "plan: bad argument (empty array or negative degree)");
    env->SetIntArrayRegion(user_begin, 0, n_parts + 1, reinterpret_cast<const jint*>(ub.get()));
    env->SetIntArrayRegion(item_part, 0, ni, reinterpret_cast<const jint*>(ip.get()));
}

JNIEXPORT void JNICALL Java_MatrixFactorizationSGD_nativeSetItemPartition(JNIEnv* env, jclass, jlong h, jintArray item_part) {
    int32_t items = 0;
    if (!item_part || mfsgd_get_dims(H(h), nullptr, &items, nullptr) != MFSGD_OK || env->GetArrayLength(item_part) != items)
        return throw_new(env, "java/lang/IllegalArgumentException", "itemPart must have one entry per item");
    auto ip = alloc<int32_t>(env, (size_t)items);
    if (!ip) return;
    env->GetIntArrayRegion(item_part, 0, items, reinterpret_cast<jint*>(ip.get()));
    if (env->ExceptionCheck()) return;
    throw_status(env, H(h), mfsgd_set_item_partition(H(h), ip.get()));
}

JNIEXPORT void JNICALL Java_MatrixFactorizationSGD_nativeInitPOffset(JNIEnv* env, jclass, jlong h, jlong seed, jlong user_offset) {
    throw_status(env, H(h), mfsgd_init_p_offset(H(h), seed, user_offset));
}

JNIEXPORT void JNICALL Java_MatrixFactorizationSGD_nativeGetUserFactors(JNIEnv* env, jclass, jlong h, jfloatArray p) {
    int32_t users = 0, k = 0;
    if (!p || mfsgd_get_dims(H(h), &users, nullptr, &k) != MFSGD_OK || (jlong)env->GetArrayLength(p) != (jlong)users * k)
        return throw_new(env, "java/lang/IllegalArgumentException", "userFactors: P must be users x k");
    const size_t np = (size_t)users * k;
    auto cp = alloc<float>(env, np);
    if (!cp) return;
    const int rc = mfsgd_get_factors(H(h), cp.get(), nullptr);
    if (rc == MFSGD_OK) env->SetFloatArrayRegion(p, 0, (jsize)np, cp.get());
    throw_status(env, H(h), rc);
}

JNIEXPORT jlong JNICALL Java_MatrixFactorizationSGD_nativeDsgdCreate(JNIEnv* env, jclass, jlong h, jint rank, jint world,
                                                                     jbyteArray id) {
    if (!id || env->GetArrayLength(id) != MFSGD_DSGD_ID_BYTES) {
        throw_new(env, "java/lang/IllegalArgumentException", "id must be the 128 bytes of distributedId()");
        return 0;
    }
    signed char raw[MFSGD_DSGD_ID_BYTES];
    env->GetByteArrayRegion(id, 0, MFSGD_DSGD_ID_BYTES, raw);
    if (env->ExceptionCheck()) return 0;
    mfsgd_dsgd* d = nullptr;
    throw_dsgd(env, nullptr, mfsgd_dsgd_create(H(h), rank, world, raw, &d));  // collective: ncclCommInitRank
    return reinterpret_cast<jlong>(d);
}

JNIEXPORT void JNICALL Java_MatrixFactorizationSGD_nativeDsgdDestroy(JNIEnv*, jclass, jlong d) { mfsgd_dsgd_destroy(D(d)); }

JNIEXPORT void JNICALL Java_MatrixFactorizationSGD_nativeDsgdInitQ(JNIEnv* env, jclass, jlong d, jlong seed, jlong users_total) {
    throw_dsgd(env, D(d), mfsgd_dsgd_init_q(D(d), seed, users_total));
}

JNIEXPORT void JNICALL Java_MatrixFactorizationSGD_nativeDsgdTrain(JNIEnv* env, jclass, jlong d, jint epochs, jdoubleArray rmse) {
    if (epochs < 0 || !rmse || env->GetArrayLength(rmse) < epochs)
        return throw_new(env, "java/lang/IllegalArgumentException", "rmse array shorter than epochs");
    auto tmp = alloc<double>(env, (size_t)epochs);
    if (!tmp) return;
    const int rc = mfsgd_dsgd_train(D(d), epochs, tmp.get());
    if (rc == MFSGD_OK && epochs > 0) env->SetDoubleArrayRegion(rmse, 0, epochs, tmp.get());
    throw_dsgd(env, D(d), rc);
}

JNIEXPORT jdouble JNICALL Java_MatrixFactorizationSGD_nativeDsgdRmse(JNIEnv* env, jclass, jlong d) {
    double out = 0.0;
    throw_dsgd(env, D(d), mfsgd_dsgd_rmse(D(d), &out));
    return out;
}

// part_rows[0] = partition id of slot j, part_rows[1] = its rows; block (nullable) receives rows x k floats
JNIEXPORT void JNICALL Java_MatrixFactorizationSGD_nativeDsgdGetQ(JNIEnv* env, jclass, jlong d, jint slot, jintArray part_rows,
                                                                  jfloatArray block) {
    if (!part_rows || env->GetArrayLength(part_rows) < 2) return throw_new(env, "java/lang/IllegalArgumentException", "partRows needs two entries");
    int32_t pr[2] = {0, 0};
    int rc = mfsgd_dsgd_get_q(D(d), slot, &pr[0], &pr[1], nullptr);
    if (rc != MFSGD_OK) return throw_dsgd(env, D(d), rc);
    env->SetIntArrayRegion(part_rows, 0, 2, reinterpret_cast<const jint*>(pr));
    if (!block) return;
    const jsize have = env->GetArrayLength(block);
    if (pr[1] > 0 && have % pr[1] != 0) return throw_new(env, "java/lang/IllegalArgumentException", "block must be rows x k");
    auto tmp = alloc<float>(env, (size_t)have);
    if (!tmp) return;
    rc = mfsgd_dsgd_get_q(D(d), slot, &pr[0], &pr[1], tmp.get());  // writes rows x k: the caller sized the array from k
    if (rc == MFSGD_OK && have > 0) env->SetFloatArrayRegion(block, 0, have, tmp.get());
    throw_dsgd(env, D(d), rc);
}

}  // extern "C"
