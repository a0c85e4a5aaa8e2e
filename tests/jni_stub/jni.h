/*
 * DECLARATION-ONLY STUB of the handful of JNI names jni/mfsgd_jni.cpp uses, transcribed from the
 * Java Native Interface specification (types, the JNIEnv member functions' signatures).  There is
 * no JDK in this image, so the real <jni.h> does not exist here; this file lets the test-suite
 * check that the shim is well-formed C++ against those signatures (hipcc -fsyntax-only).  It is
 * TEST INFRASTRUCTURE: nothing links against it, it implements nothing, and a shim that passes
 * this check has still never run under a JVM.
 */
#ifndef MFSGD_TEST_JNI_STUB_H
#define MFSGD_TEST_JNI_STUB_H
#include <cstdint>
typedef int32_t jint;
typedef int64_t jlong;
typedef float jfloat;
typedef double jdouble;
typedef jint jsize;
typedef unsigned char jboolean;
typedef signed char jbyte;
class _jobject {};
class _jclass : public _jobject {};
class _jarray : public _jobject {};
class _jintArray : public _jarray {};
class _jfloatArray : public _jarray {};
class _jdoubleArray : public _jarray {};
class _jbyteArray : public _jarray {};
class _jlongArray : public _jarray {};
typedef _jobject* jobject;
typedef _jclass* jclass;
typedef _jarray* jarray;
typedef _jintArray* jintArray;
typedef _jfloatArray* jfloatArray;
typedef _jdoubleArray* jdoubleArray;
typedef _jbyteArray* jbyteArray;
typedef _jlongArray* jlongArray;
#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_ABORT 2
struct JNIEnv {
    jclass FindClass(const char* name);
    jint ThrowNew(jclass cls, const char* msg);
    jboolean ExceptionCheck();
    jsize GetArrayLength(jarray a);
    void* GetPrimitiveArrayCritical(jarray a, jboolean* is_copy);
    void ReleasePrimitiveArrayCritical(jarray a, void* carray, jint mode);
    void GetIntArrayRegion(jintArray a, jsize start, jsize len, jint* buf);
    void GetFloatArrayRegion(jfloatArray a, jsize start, jsize len, jfloat* buf);
    void SetIntArrayRegion(jintArray a, jsize start, jsize len, const jint* buf);
    void SetFloatArrayRegion(jfloatArray a, jsize start, jsize len, const jfloat* buf);
    void SetDoubleArrayRegion(jdoubleArray a, jsize start, jsize len, const jdouble* buf);
    jbyteArray NewByteArray(jsize len);
    void GetByteArrayRegion(jbyteArray a, jsize start, jsize len, jbyte* buf);
    void SetByteArrayRegion(jbyteArray a, jsize start, jsize len, const jbyte* buf);
    void GetLongArrayRegion(jlongArray a, jsize start, jsize len, jlong* buf);
};
#endif
