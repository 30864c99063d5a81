/* TEST INFRASTRUCTURE -- NOT a JDK header.  This image has no JDK, so integration/bvgraph_hip_jni.c (OUR shim, the file a webgraph-big maintainer would compile against a real
 * <jni.h>) had never been through a compiler.  This file declares, from the published JNI specification (Java Native Interface Specification, chapter 4 "JNI Functions": the
 * C form of JNIEnv is a pointer to a table of function pointers, every function takes the JNIEnv* first), exactly the types and the 13 functions the shim uses, so that
 * `gcc -fsyntax-only -Wall -Werror` can type-check it (tests/test_integration_sources.py).  The function table here is NOT laid out like the real one: nothing compiled against this
 * header may ever be linked or run. */
#ifndef BVG_TEST_JNI_STUB_H
#define BVG_TEST_JNI_STUB_H
#include <stdint.h>

typedef int32_t jint;
typedef int64_t jlong;
typedef uint8_t jboolean;
typedef jint jsize;
struct _jobject;
typedef struct _jobject* jobject;
typedef jobject jclass;
typedef jobject jstring;
typedef jobject jarray;
typedef jarray jintArray;
typedef jarray jlongArray;

#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_ABORT 2
#define JNI_FALSE 0
#define JNI_TRUE 1

struct JNINativeInterface_;
typedef const struct JNINativeInterface_* JNIEnv;
struct JNINativeInterface_ {
    jclass (*FindClass)(JNIEnv*, const char* name);
    jint (*ThrowNew)(JNIEnv*, jclass clazz, const char* message);
    jsize (*GetArrayLength)(JNIEnv*, jarray array);
    const char* (*GetStringUTFChars)(JNIEnv*, jstring string, jboolean* isCopy);
    void (*ReleaseStringUTFChars)(JNIEnv*, jstring string, const char* utf);
    jlongArray (*NewLongArray)(JNIEnv*, jsize length);
    jlong* (*GetLongArrayElements)(JNIEnv*, jlongArray array, jboolean* isCopy);
    void (*ReleaseLongArrayElements)(JNIEnv*, jlongArray array, jlong* elems, jint mode);
    void (*SetIntArrayRegion)(JNIEnv*, jintArray array, jsize start, jsize len, const jint* buf);
    void (*SetLongArrayRegion)(JNIEnv*, jlongArray array, jsize start, jsize len, const jlong* buf);
    jobject (*NewDirectByteBuffer)(JNIEnv*, void* address, jlong capacity);
    void* (*GetDirectBufferAddress)(JNIEnv*, jobject buf);
    jlong (*GetDirectBufferCapacity)(JNIEnv*, jobject buf);
};
#endif
