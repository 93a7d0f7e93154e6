"""integration/jni/maelsim_jni.c is the binding a maintainer would add on the JVM side.  There is no JDK in this
image, so the shim cannot be built for real; this keeps it honest against include/maelsim.h anyway: it must compile (syntax and
types) against the header with a minimal stand-in for <jni.h> that declares only what the shim uses."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

FAKE_JNI = r"""
#ifndef FAKE_JNI_H
#define FAKE_JNI_H
#include <stdint.h>
#define JNIEXPORT
#define JNICALL
typedef int32_t jint; typedef int64_t jlong; typedef int32_t jsize;
typedef void *jobject; typedef jobject jclass; typedef jobject jstring; typedef jobject jarray; typedef jarray jintArray;
typedef jarray jlongArray; typedef jarray jobjectArray;
struct JNINativeInterface_;
typedef const struct JNINativeInterface_ *JNIEnv;
struct JNINativeInterface_ {
  jclass (*FindClass)(JNIEnv *, const char *);
  jint (*ThrowNew)(JNIEnv *, jclass, const char *);
  void (*GetIntArrayRegion)(JNIEnv *, jintArray, jsize, jsize, jint *);
  jobjectArray (*NewObjectArray)(JNIEnv *, jsize, jclass, jobject);
  void (*SetObjectArrayElement)(JNIEnv *, jobjectArray, jsize, jobject);
  jobject (*NewDirectByteBuffer)(JNIEnv *, void *, jlong);
  jlongArray (*NewLongArray)(JNIEnv *, jsize);
  void (*SetLongArrayRegion)(JNIEnv *, jlongArray, jsize, jsize, const jlong *);
  jstring (*NewStringUTF)(JNIEnv *, const char *);
};
#endif
"""


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_jni_shim_compiles_against_the_header(tmp_path):
    (tmp_path / "jni.h").write_text(FAKE_JNI)
    r = subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-fsyntax-only", "-I", str(tmp_path), "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "integration", "jni", "maelsim_jni.c")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
