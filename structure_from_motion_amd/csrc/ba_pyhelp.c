/* Host-side marshalling helper of the drop-in module (CPython extension, plain C; no GPU, no numerics).
 *
 * The reference's driver hands bundle adjustment its state as Python containers: all_points_3D, a list of
 * [x, y, z, r, g, b] lists, and per image a list of cv2.KeyPoint objects plus a list of point references
 * (Reconstruction.py:41-47, 59-60; read by BundleAdjustment.py:243-272).  Walking 20 000 point rows and 80 000
 * keypoints object by object is what a Temple-sized final_ba call spends most of its time on once the solve itself
 * takes a millisecond.  These three functions do exactly those walks in C, straight into NumPy buffers; they return
 * None whenever a container is not shaped as expected, and the caller then takes its generic Python path (same
 * result).  Semantics are those of BundleAdjustment.py's own loops of this package, which the tests compare.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>

static int as_double(PyObject *o, double *out) {
    if (PyFloat_CheckExact(o)) {
        *out = PyFloat_AS_DOUBLE(o);
        return 1;
    }
    const double v = PyFloat_AsDouble(o); /* ints, numpy scalars */
    if (v == -1.0 && PyErr_Occurred()) {
        PyErr_Clear();
        return 0;
    }
    *out = v;
    return 1;
}

/* rows_to_array(rows, out) -> width or None.  rows: list of equally long lists / tuples of numbers; out: writable
 * buffer of float64, at least len(rows) * width elements, filled row-major. */
static PyObject *rows_to_array(PyObject *self, PyObject *args) {
    PyObject *rows;
    Py_buffer out;
    if (!PyArg_ParseTuple(args, "Ow*", &rows, &out)) return NULL;
    PyObject *ret = NULL;
    if (!PyList_CheckExact(rows) || PyList_GET_SIZE(rows) == 0) goto none;
    {
        const Py_ssize_t n = PyList_GET_SIZE(rows);
        PyObject *first = PyList_GET_ITEM(rows, 0);
        Py_ssize_t w;
        if (PyList_CheckExact(first)) w = PyList_GET_SIZE(first);
        else if (PyTuple_CheckExact(first)) w = PyTuple_GET_SIZE(first);
        else goto none;
        if (w <= 0 || (Py_ssize_t)(out.len / (Py_ssize_t)sizeof(double)) < n * w) goto none;
        double *d = (double *)out.buf;
        for (Py_ssize_t i = 0; i < n; ++i) {
            PyObject *row = PyList_GET_ITEM(rows, i);
            PyObject **items;
            if (PyList_CheckExact(row) && PyList_GET_SIZE(row) == w) items = ((PyListObject *)row)->ob_item;
            else if (PyTuple_CheckExact(row) && PyTuple_GET_SIZE(row) == w) items = ((PyTupleObject *)row)->ob_item;
            else goto none;
            for (Py_ssize_t k = 0; k < w; ++k)
                if (!as_double(items[k], &d[i * w + k])) goto none;
        }
        ret = PyLong_FromSsize_t(w);
    }
    goto done;
none:
    ret = Py_None;
    Py_INCREF(ret);
done:
    PyBuffer_Release(&out);
    return ret;
}

/* track_entries(ref_array, n_kp, n_points) -> bytes of int64 pairs (keypoint index, point index) or None.
 * The reference's rule (BundleAdjustment.py:260-264): entry != -1, keypoint exists, a Python int, 0 <= entry < n_points.
 * None: ref_array is not a list (the caller's generic path decides). */
static PyObject *track_entries(PyObject *self, PyObject *args) {
    PyObject *ref;
    Py_ssize_t n_kp, n_points;
    if (!PyArg_ParseTuple(args, "Onn", &ref, &n_kp, &n_points)) return NULL;
    if (!PyList_CheckExact(ref)) Py_RETURN_NONE;
    const Py_ssize_t n = PyList_GET_SIZE(ref);
    const Py_ssize_t lim = n < n_kp ? n : n_kp;
    PyObject *bytes = PyBytes_FromStringAndSize(NULL, (lim > 0 ? lim : 0) * 2 * (Py_ssize_t)sizeof(int64_t));
    if (!bytes) return NULL;
    int64_t *o = (int64_t *)PyBytes_AS_STRING(bytes);
    Py_ssize_t m = 0;
    for (Py_ssize_t k = 0; k < lim; ++k) {
        PyObject *v = PyList_GET_ITEM(ref, k);
        if (!PyLong_Check(v) || PyBool_Check(v)) {
            if (PyBool_Check(v)) {   /* isinstance(True, int) holds in Python: same rule */
                const long b = v == Py_True;
                if (b != -1 && b >= 0 && b < n_points) { o[2 * m] = k; o[2 * m + 1] = b; ++m; }
            }
            continue;
        }
        int overflow = 0;
        const long long x = PyLong_AsLongLongAndOverflow(v, &overflow);
        if (overflow || x == -1 || x < 0 || x >= (long long)n_points) continue;
        o[2 * m] = (int64_t)k;
        o[2 * m + 1] = (int64_t)x;
        ++m;
    }
    if (_PyBytes_Resize(&bytes, m * 2 * (Py_ssize_t)sizeof(int64_t)) < 0) return NULL;
    return bytes;
}

/* keypoints_xy(kp, kidx, out) -> True or None.  kp: list of objects with a .pt attribute (a pair of numbers,
 * cv2.KeyPoint.pt, BundleAdjustment.py:269); kidx: buffer of int64 indices into kp; out: writable float64 buffer of
 * 2 * len(kidx) elements. */
static PyObject *keypoints_xy(PyObject *self, PyObject *args) {
    PyObject *kp;
    Py_buffer idx, out;
    if (!PyArg_ParseTuple(args, "Oy*w*", &kp, &idx, &out)) return NULL;
    PyObject *ret = NULL;
    static PyObject *pt_name = NULL;
    if (!pt_name) pt_name = PyUnicode_InternFromString("pt");
    if (!PyList_CheckExact(kp) || !pt_name) goto none;
    {
        const Py_ssize_t n = idx.len / (Py_ssize_t)sizeof(int64_t), n_kp = PyList_GET_SIZE(kp);
        if ((Py_ssize_t)(out.len / (Py_ssize_t)sizeof(double)) < 2 * n) goto none;
        const int64_t *ix = (const int64_t *)idx.buf;
        double *d = (double *)out.buf;
        for (Py_ssize_t i = 0; i < n; ++i) {
            if (ix[i] < 0 || ix[i] >= n_kp) goto none;
            PyObject *pt = PyObject_GetAttr(PyList_GET_ITEM(kp, ix[i]), pt_name);
            if (!pt) {
                PyErr_Clear();
                goto none;
            }
            int ok = 0;
            if (PyTuple_CheckExact(pt) && PyTuple_GET_SIZE(pt) == 2)
                ok = as_double(PyTuple_GET_ITEM(pt, 0), &d[2 * i]) && as_double(PyTuple_GET_ITEM(pt, 1), &d[2 * i + 1]);
            Py_DECREF(pt);
            if (!ok) goto none;
        }
        ret = Py_True;
        Py_INCREF(ret);
    }
    goto done;
none:
    ret = Py_None;
    Py_INCREF(ret);
done:
    PyBuffer_Release(&idx);
    PyBuffer_Release(&out);
    return ret;
}

/* rows_with_xyz(rows, xyz) -> new list of rows or None.  What the reference returns for the points (BundleAdjustment.py:
 * 96-97): [x, y, z] from the optimised parameters followed by the ORIGINAL colour entries of each row (the very objects,
 * as the reference's `+ original colours` does).  rows: list of lists / tuples of at least 3 entries; xyz: float64 buffer
 * of 3 * len(rows).  None: an unexpected container (the caller's Python path handles it). */
static PyObject *rows_with_xyz(PyObject *self, PyObject *args) {
    PyObject *rows;
    Py_buffer xyz;
    if (!PyArg_ParseTuple(args, "Oy*", &rows, &xyz)) return NULL;
    PyObject *ret = NULL;
    if (!PyList_CheckExact(rows)) goto none;
    {
        const Py_ssize_t n = PyList_GET_SIZE(rows);
        if ((Py_ssize_t)(xyz.len / (Py_ssize_t)sizeof(double)) < 3 * n) goto none;
        const double *d = (const double *)xyz.buf;
        for (Py_ssize_t i = 0; i < n; ++i) {   /* shapes first: nothing is built for a container the caller must handle */
            PyObject *row = PyList_GET_ITEM(rows, i);
            if (!(PyList_CheckExact(row) && PyList_GET_SIZE(row) >= 3) && !(PyTuple_CheckExact(row) && PyTuple_GET_SIZE(row) >= 3)) goto none;
        }
        ret = PyList_New(n);
        if (!ret) goto done;
        for (Py_ssize_t i = 0; i < n; ++i) {
            PyObject *row = PyList_GET_ITEM(rows, i);
            const int is_list = PyList_CheckExact(row);
            const Py_ssize_t w = is_list ? PyList_GET_SIZE(row) : PyTuple_GET_SIZE(row);
            PyObject **items = is_list ? ((PyListObject *)row)->ob_item : ((PyTupleObject *)row)->ob_item;
            const Py_ssize_t wn = w < 6 ? w : 6;   /* the reference keeps entries 3, 4, 5 */
            PyObject *nr = PyList_New(wn);
            if (!nr) { Py_CLEAR(ret); goto done; }
            for (Py_ssize_t k = 0; k < 3; ++k) {
                PyObject *f = PyFloat_FromDouble(d[3 * i + k]);
                if (!f) { Py_DECREF(nr); Py_CLEAR(ret); goto done; }
                PyList_SET_ITEM(nr, k, f);
            }
            for (Py_ssize_t k = 3; k < wn; ++k) {
                Py_INCREF(items[k]);
                PyList_SET_ITEM(nr, k, items[k]);
            }
            PyList_SET_ITEM(ret, i, nr);
        }
    }
    goto done;
none:
    ret = Py_None;
    Py_INCREF(ret);
done:
    PyBuffer_Release(&xyz);
    return ret;
}

static PyMethodDef methods[] = {
    {"rows_with_xyz", rows_with_xyz, METH_VARARGS, "new point rows: [x, y, z] from a float64 buffer + each row's own entries 3..5, or None"},
    {"rows_to_array", rows_to_array, METH_VARARGS, "list of equally long number rows -> float64 buffer; width or None"},
    {"track_entries", track_entries, METH_VARARGS, "valid (keypoint, point) index pairs of one image as int64 bytes, or None"},
    {"keypoints_xy", keypoints_xy, METH_VARARGS, "kp[k].pt for the given indices -> float64 buffer; True or None"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_ba_pyhelp", "marshalling walks of the BundleAdjustment drop-in in C", -1, methods};

PyMODINIT_FUNC PyInit__ba_pyhelp(void) { return PyModule_Create(&moddef); }
