// CriCodecs_ext.cpp -- CPython extension module `CriCodecs` with the reference's codec method table
// (/root/reference/CriCodecs/CriCodecs.cpp:8-17), implemented as a thin binding over the C ABI of
// libcricodecs_hip.so (include/cricodecs_hip.h), which it dlopen()s next to itself.  This is the host side
// "in the reference's own language": a PyCriCodecs checkout can drop this module in place of its own CriCodecs
// extension and keep PyCriCodecs/adx.py and hca.py unchanged.
//
// Differences from the reference wrappers (adx.cpp:517-558, hca.cpp:3271-3489), all fixes of undefined behaviour
// listed in SURVEY.md section 9: arguments are parsed into correctly typed variables, input lengths are passed down,
// the GIL is released around the device call, HcaCrypt works on a private copy, error state is per call.
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <dlfcn.h>
#include <stdint.h>
#include <string>

namespace {

typedef int (*fn_adx_decode)(const uint8_t*, size_t, uint8_t**, size_t*);
typedef int (*fn_adx_encode)(const uint8_t*, size_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, int, uint8_t**, size_t*);
typedef int (*fn_hca_decode)(const uint8_t*, size_t, uint32_t, uint64_t, uint16_t, uint8_t**, size_t*);
typedef int (*fn_hca_encode)(const uint8_t*, size_t, uint32_t, uint32_t, uint8_t**, size_t*);
typedef int (*fn_hca_crypt)(uint8_t*, size_t, uint32_t, uint32_t, uint32_t, uint64_t, uint16_t);
typedef void (*fn_free)(void*);
typedef const char* (*fn_strerror)(int);

struct Api {
    void* handle = nullptr;
    fn_adx_decode adx_decode; fn_adx_encode adx_encode; fn_hca_decode hca_decode; fn_hca_encode hca_encode; fn_hca_crypt hca_crypt;
    fn_free free_; fn_strerror strerror_;
} api;

bool load_api() {
    if (api.handle) return true;
    Dl_info info;
    std::string dir = ".";
    if (dladdr((void*)&load_api, &info) && info.dli_fname) { dir = info.dli_fname; size_t p = dir.rfind('/'); dir = p == std::string::npos ? "." : dir.substr(0, p); }
    const char* env = getenv("CRICODECS_HIP_LIB");
    std::string path = env ? env : dir + "/libcricodecs_hip.so";
    void* h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!h) { PyErr_Format(PyExc_ImportError, "cannot load %s: %s", path.c_str(), dlerror()); return false; }
#define SYM(field, type, name) api.field = (type)dlsym(h, name); if (!api.field) { PyErr_Format(PyExc_ImportError, "%s lacks %s", path.c_str(), name); dlclose(h); return false; }
    SYM(adx_decode, fn_adx_decode, "cri_adx_decode") SYM(adx_encode, fn_adx_encode, "cri_adx_encode")
    SYM(hca_decode, fn_hca_decode, "cri_hca_decode") SYM(hca_encode, fn_hca_encode, "cri_hca_encode")
    SYM(hca_crypt, fn_hca_crypt, "cri_hca_crypt") SYM(free_, fn_free, "cri_free") SYM(strerror_, fn_strerror, "cri_strerror")
#undef SYM
    api.handle = h;
    return true;
}

// same exception types / messages as PyAdxSetError (adx.cpp:32-38), PyPCMSetError (pcm.cpp:35-38), py_codec_err (hca.cpp:3252-3268)
PyObject* raise_code(int code) {
    const char* msg = api.strerror_(code);
    if (code == -3) PyErr_SetString(PyExc_NotImplementedError, msg);
    else if ((code <= -1 && code >= -18) || (code <= -101 && code >= -110) || (code <= -201 && code >= -216)) PyErr_SetString(PyExc_ValueError, msg);
    else PyErr_SetString(PyExc_RuntimeError, msg);
    return nullptr;
}

PyObject* take(int rc, uint8_t* out, size_t n) {
    if (rc) return raise_code(rc);
    PyObject* r = PyBytes_FromStringAndSize((const char*)out, (Py_ssize_t)n);
    api.free_(out);
    return r;
}

PyObject* AdxDecode(PyObject*, PyObject* arg) {
    if (!load_api()) return nullptr;
    char* data; Py_ssize_t len;
    if (PyBytes_AsStringAndSize(arg, &data, &len) < 0) return nullptr;
    uint8_t* out = nullptr; size_t n = 0; int rc;
    Py_BEGIN_ALLOW_THREADS rc = api.adx_decode((const uint8_t*)data, (size_t)len, &out, &n); Py_END_ALLOW_THREADS
    return take(rc, out, n);
}

PyObject* AdxEncode(PyObject*, PyObject* args) {
    if (!load_api()) return nullptr;
    const char* data; Py_ssize_t len; unsigned int bitdepth, blocksize, encoding, highpass, filter, version; int force;
    if (!PyArg_ParseTuple(args, "y#IIIIIIp", &data, &len, &bitdepth, &blocksize, &encoding, &highpass, &filter, &version, &force)) return nullptr;
    uint8_t* out = nullptr; size_t n = 0; int rc;
    Py_BEGIN_ALLOW_THREADS rc = api.adx_encode((const uint8_t*)data, (size_t)len, bitdepth, blocksize, encoding, highpass, filter, version, force, &out, &n); Py_END_ALLOW_THREADS
    return take(rc, out, n);
}

PyObject* HcaDecode(PyObject*, PyObject* args) {
    if (!load_api()) return nullptr;
    const char* data; Py_ssize_t len; unsigned int header_size; unsigned long long key; unsigned short subkey;
    if (!PyArg_ParseTuple(args, "y#IKH", &data, &len, &header_size, &key, &subkey)) return nullptr;
    uint8_t* out = nullptr; size_t n = 0; int rc;
    Py_BEGIN_ALLOW_THREADS rc = api.hca_decode((const uint8_t*)data, (size_t)len, header_size, key, subkey, &out, &n); Py_END_ALLOW_THREADS
    return take(rc, out, n);
}

PyObject* HcaEncode(PyObject*, PyObject* args) {
    if (!load_api()) return nullptr;
    Py_buffer view; unsigned int force, quality;
    if (!PyArg_ParseTuple(args, "y*II", &view, &force, &quality)) return nullptr;
    uint8_t* out = nullptr; size_t n = 0; int rc;
    Py_BEGIN_ALLOW_THREADS rc = api.hca_encode((const uint8_t*)view.buf, (size_t)view.len, force, quality, &out, &n); Py_END_ALLOW_THREADS
    PyBuffer_Release(&view);
    return take(rc, out, n);
}

PyObject* HcaCrypt(PyObject*, PyObject* args) {
    if (!load_api()) return nullptr;
    Py_buffer view; unsigned int crypt, header_size, type; unsigned long long key; unsigned short subkey;
    if (!PyArg_ParseTuple(args, "y*IIIKH", &view, &crypt, &header_size, &type, &key, &subkey)) return nullptr;
    PyObject* copy = PyBytes_FromStringAndSize((const char*)view.buf, view.len);
    PyBuffer_Release(&view);
    if (!copy) return nullptr;
    int rc;
    uint8_t* p = (uint8_t*)PyBytes_AS_STRING(copy); size_t n = (size_t)PyBytes_GET_SIZE(copy);
    Py_BEGIN_ALLOW_THREADS rc = api.hca_crypt(p, n, crypt, header_size, type, key, subkey); Py_END_ALLOW_THREADS
    if (rc) { Py_DECREF(copy); return raise_code(rc); }
    return copy;
}

PyMethodDef methods[] = {
    {"AdxDecode", (PyCFunction)AdxDecode, METH_O, nullptr},
    {"AdxEncode", (PyCFunction)AdxEncode, METH_VARARGS, nullptr},
    {"HcaDecode", (PyCFunction)HcaDecode, METH_VARARGS, nullptr},
    {"HcaEncode", (PyCFunction)HcaEncode, METH_VARARGS, nullptr},
    {"HcaCrypt", (PyCFunction)HcaCrypt, METH_VARARGS, nullptr},
    {nullptr, nullptr, 0, nullptr}};

PyModuleDef module = {PyModuleDef_HEAD_INIT, "CriCodecs", "ADX / HCA codecs on MI355X (libcricodecs_hip.so)", 0, methods};

}  // namespace

PyMODINIT_FUNC PyInit_CriCodecs() { return PyModule_Create(&module); }
