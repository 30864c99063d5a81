"""integration/: the Java class and the JNI shim a webgraph-big maintainer would add.  No JDK exists in this image, so neither has ever been built for real; what CAN be checked here is:
  * the C shim type-checks (gcc -fsyntax-only -Wall -Wextra -Werror) against tests/cpp/jni_stub/jni.h -- a declaration of the JNI types and the 13 JNI functions it uses, written from
    the published JNI specification (test infrastructure: nothing compiled against it is ever linked or run) -- and against the product's real include/bvgraph_hip.h;
  * every `native` method of HipBVGraph.java has its Java_it_unimi_dsi_big_webgraph_HipBVGraph_<name> function in the shim, with the parameter and return types the JNI type
    mapping prescribes (JNI specification, chapter 3 "JNI Types and Data Structures")."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JAVA = os.path.join(ROOT, "integration", "HipBVGraph.java")
SHIM = os.path.join(ROOT, "integration", "bvgraph_hip_jni.c")

JNI = {"void": "void", "int": "jint", "long": "jlong", "boolean": "jboolean", "String": "jstring", "long[]": "jlongArray", "int[]": "jintArray",
       "IntBuffer": "jobject", "LongBuffer": "jobject", "ByteBuffer": "jobject"}


def test_jni_shim_type_checks_against_the_jni_stub_and_the_real_header():
    out = subprocess.run(["gcc", "-std=c99", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "tests", "cpp", "jni_stub"), "-I" + os.path.join(ROOT, "include"), SHIM],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-3000:]


def test_every_native_method_has_its_jni_function_with_matching_types():
    java = open(JAVA).read()
    shim = open(SHIM).read()
    natives = re.findall(r"private static native\s+([\w\[\]]+)\s+(\w+)\(([^)]*)\)", java)
    assert len(natives) >= 16
    for ret, name, params in natives:
        jtypes = [p.strip().rsplit(" ", 1)[0].replace("final ", "").strip() for p in params.split(",") if p.strip()]
        m = re.search(r"JNIEXPORT\s+(\w+)\s+JNICALL\s+Java_it_unimi_dsi_big_webgraph_HipBVGraph_%s\(([^)]*)\)" % name, shim)
        assert m, "no JNI function for native method %s" % name
        cparams = [p.strip().rsplit(" ", 1)[0].strip() for p in m.group(2).split(",")]
        assert cparams[:2] == ["JNIEnv*", "jclass"], (name, cparams)                  # static native methods take the class
        assert m.group(1) == JNI[ret], (name, ret, m.group(1))
        assert cparams[2:] == [JNI[t] for t in jtypes], (name, jtypes, cparams)
    # and nothing in the shim that the class does not declare
    for fn in re.findall(r"Java_it_unimi_dsi_big_webgraph_HipBVGraph_(\w+)\(", shim):
        assert any(fn == n for _, n, _ in natives), fn
