"""The JVM binding is source only (no JVM in the image).  What can be checked here: the Java `native` declarations, the JNI
functions and the Scala call sites name the same methods, and the JNI shim type-checks against include/gingr_hip.h (compiled
with g++ -fsyntax-only against a minimal stand-in for <jni.h> that only declares the few types and calls the shim uses)."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JAVA = os.path.join(ROOT, "jvm", "src", "main", "java", "gingr", "hip", "GingrHipNative.java")
JNI = os.path.join(ROOT, "jvm", "native", "gingr_jni.cpp")
SCALA = os.path.join(ROOT, "jvm", "src", "main", "scala", "gingr", "api", "registration", "config", "HipCPD.scala")

MOCK_JNI = """
#pragma once
#include <cstdint>
typedef int32_t jint; typedef int32_t jsize; typedef int64_t jlong; typedef double jdouble; typedef unsigned char jboolean;
class _jobject {}; typedef _jobject *jobject; typedef jobject jclass; typedef jobject jarray; typedef jarray jdoubleArray;
typedef jarray jintArray; typedef jobject jstring;
#define JNIEXPORT
#define JNICALL
#define JNI_ABORT 2
struct JNIEnv {
    void *GetPrimitiveArrayCritical(jarray, jboolean *) { return nullptr; }
    void ReleasePrimitiveArrayCritical(jarray, void *, jint) {}
    jsize GetArrayLength(jarray) { return 0; }
    jstring NewStringUTF(const char *) { return nullptr; }
};
"""


def test_java_jni_and_scala_name_the_same_methods():
    java, jni, scala = open(JAVA).read(), open(JNI).read(), open(SCALA).read()
    natives = set(re.findall(r"public static native \S+ (\w+)\(", java))
    shims = set(re.findall(r"JFN\([^,]+,\s*(\w+)\)", jni)) - {"name"}
    assert natives == shims, (sorted(natives - shims), sorted(shims - natives))
    used = set(re.findall(r"GingrHipNative\.(\w+)", scala))
    assert used <= natives, sorted(used - natives)
    assert len(natives) >= 40


def test_jni_shim_type_checks_against_the_header(tmp_path):
    (tmp_path / "jni.h").write_text(MOCK_JNI)
    out = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I", str(tmp_path), "-I", os.path.join(ROOT, "include"), JNI],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[:2000]
