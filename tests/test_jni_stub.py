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
typedef int32_t jint; typedef int64_t jlong; typedef int32_t jsize; typedef signed char jbyte;
typedef void *jobject; typedef jobject jclass; typedef jobject jstring; typedef jobject jarray; typedef jarray jintArray;
typedef jarray jlongArray; typedef jarray jobjectArray; typedef jarray jbyteArray;
struct JNINativeInterface_;
typedef const struct JNINativeInterface_ *JNIEnv;
struct JNINativeInterface_ {
  jclass (*FindClass)(JNIEnv *, const char *);
  jint (*ThrowNew)(JNIEnv *, jclass, const char *);
  jsize (*GetArrayLength)(JNIEnv *, jarray);
  jbyteArray (*NewByteArray)(JNIEnv *, jsize);
  void (*GetByteArrayRegion)(JNIEnv *, jbyteArray, jsize, jsize, jbyte *);
  void (*SetByteArrayRegion)(JNIEnv *, jbyteArray, jsize, jsize, const jbyte *);
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


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_plain_c_caller_compiles(tmp_path):
    """integration/c/smoke.c — the library used from C through dlopen, no binding layer at all (run on the GPU by tests/test_abi_c_smoke_gpu.py)"""
    r = subprocess.run(["gcc", "-std=gnu11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "integration", "c", "smoke.c"),
                        "-ldl", "-o", str(tmp_path / "smoke")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_clojure_stub_config_offsets_match_the_header():
    """integration/clojure/maelstrom/gpu.clj writes msim_config fields at byte offsets: they must be the header's"""
    import ctypes as C
    import re
    from maelstrom_amd import _abi as A
    src = open(os.path.join(ROOT, "integration", "clojure", "maelstrom", "gpu.clj")).read()
    m = re.search(r"\(def config-offsets\s+\{(.*?)\}\)", src, re.S)
    assert m, "config-offsets map not found"
    offs = {k: int(v) for k, v in re.findall(r":([a-z0-9-]+)\s+(\d+)", m.group(1))}
    for name, _ in A.Config._fields_:
        assert offs[name.replace("_", "-")] == getattr(A.Config, name).offset, name
    assert int(re.search(r"\(def config-size (\d+)\)", src).group(1)) == C.sizeof(A.Config)
