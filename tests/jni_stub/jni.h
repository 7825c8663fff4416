/*
 * TEST SCAFFOLDING ONLY -- not a JDK header and not part of the product.
 *
 * The image this repository is developed in has no JDK, so java/src/main/c/cds_hip_jni.c has never met a compiler.  This file
 * declares, from the JNI specification (types of section "Primitive Types" / "Reference Types", the JNIEnv function table entries
 * the shim uses), just enough for `gcc -fsyntax-only -Wall -Werror` to PARSE and TYPE-CHECK that file
 * (tests/test_java_sources.py::test_jni_shim_parses_and_type_checks).  Nothing is ever linked or run against it: a real build
 * uses the JDK's own <jni.h> (java/Makefile), whose function table has the same member names and signatures.
 */
#ifndef CDS_TEST_JNI_STUB_H
#define CDS_TEST_JNI_STUB_H
#include <stdint.h>

#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_ABORT 2
#define JNI_COMMIT 1

typedef int32_t jint;
typedef int64_t jlong;
typedef int8_t jbyte;
typedef uint8_t jboolean;
typedef uint16_t jchar;
typedef int16_t jshort;
typedef float jfloat;
typedef double jdouble;
typedef jint jsize;

struct _jobject;
typedef struct _jobject *jobject;
typedef jobject jclass;
typedef jobject jthrowable;
typedef jobject jstring;
typedef jobject jarray;
typedef jarray jbooleanArray;
typedef jarray jbyteArray;
typedef jarray jcharArray;
typedef jarray jshortArray;
typedef jarray jintArray;
typedef jarray jlongArray;
typedef jarray jfloatArray;
typedef jarray jdoubleArray;
typedef jarray jobjectArray;

struct JNINativeInterface_;
typedef const struct JNINativeInterface_ *JNIEnv;

struct JNINativeInterface_ {
    jclass (*FindClass)(JNIEnv *env, const char *name);
    jint (*ThrowNew)(JNIEnv *env, jclass clazz, const char *msg);
    void (*DeleteLocalRef)(JNIEnv *env, jobject obj);
    jsize (*GetArrayLength)(JNIEnv *env, jarray array);
    jobject (*GetObjectArrayElement)(JNIEnv *env, jobjectArray array, jsize index);
    const char *(*GetStringUTFChars)(JNIEnv *env, jstring str, jboolean *isCopy);
    void (*ReleaseStringUTFChars)(JNIEnv *env, jstring str, const char *chars);
    jintArray (*NewIntArray)(JNIEnv *env, jsize len);
    jlongArray (*NewLongArray)(JNIEnv *env, jsize len);
    jbyte *(*GetByteArrayElements)(JNIEnv *env, jbyteArray array, jboolean *isCopy);
    jshort *(*GetShortArrayElements)(JNIEnv *env, jshortArray array, jboolean *isCopy);
    jint *(*GetIntArrayElements)(JNIEnv *env, jintArray array, jboolean *isCopy);
    jlong *(*GetLongArrayElements)(JNIEnv *env, jlongArray array, jboolean *isCopy);
    void (*ReleaseByteArrayElements)(JNIEnv *env, jbyteArray array, jbyte *elems, jint mode);
    void (*ReleaseShortArrayElements)(JNIEnv *env, jshortArray array, jshort *elems, jint mode);
    void (*ReleaseIntArrayElements)(JNIEnv *env, jintArray array, jint *elems, jint mode);
    void (*ReleaseLongArrayElements)(JNIEnv *env, jlongArray array, jlong *elems, jint mode);
    void (*GetIntArrayRegion)(JNIEnv *env, jintArray array, jsize start, jsize len, jint *buf);
    void (*SetIntArrayRegion)(JNIEnv *env, jintArray array, jsize start, jsize len, const jint *buf);
    void (*SetLongArrayRegion)(JNIEnv *env, jlongArray array, jsize start, jsize len, const jlong *buf);
    void *(*GetDirectBufferAddress)(JNIEnv *env, jobject buf);
    jlong (*GetDirectBufferCapacity)(JNIEnv *env, jobject buf);
};
#endif
