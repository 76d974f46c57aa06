/* Host-side helper of the Python facade (nlsh_amd/indexer.py): the reference's Indexer.query returns List[List[int]]
 * (nlsh/indexer.py:90-96: `indexes.tolist()` per query), i.e. 10^5 Python ints in 10^4 lists per 10^4-query batch, and building
 * them is the longest stage of a query() call once the device side takes 0.37 ms.  numpy's `ndarray.tolist()` goes through its
 * generic per-element getitem; this is the same construction as one tight loop (PyList_New + PyLong_FromLong), ~20 % faster.
 * Plain CPython C API, no GPU code; optional: the facade falls back to `ndarray.tolist()` when the module is not built.
 * Same result, element for element (tests/test_host_cpu.py).
 *
 * OPT-IN (fourth argument, default 0; `Indexer.untracked_results`): the inner lists are handed out UNTRACKED by the cyclic garbage
 * collector (PyObject_GC_UnTrack).  A list of ints cannot be part of a reference cycle, and CPython itself untracks tuples and dicts of
 * atomic objects for the same reason; 10^4 tracked young lists per call are what makes the collector's next generation-0 pass cost
 * 0.26 ms, and untracked they are invisible to it.  But CPython never untracks LISTS: a caller that later builds a
 * reference cycle through such a row (row.append(obj_that_references_row)) would leak it, so by default the rows are ordinary tracked
 * lists, exactly what `.tolist()` returns in the reference (nlsh/indexer.py:91); r03 had this on whenever the helper was built. */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>

/* rows_to_lists(buffer of int32 [Q, k] C-contiguous, Q, k, untrack=0) -> [[int] * k] * Q */
static PyObject *rows_to_lists(PyObject *self, PyObject *args) {
    Py_buffer view;
    Py_ssize_t Q, k;
    int untrack = 0;
    (void)self;
    if (!PyArg_ParseTuple(args, "y*nn|p", &view, &Q, &k, &untrack)) return NULL;
    if (Q < 0 || k < 0 || view.len < Q * k * 4) {
        PyBuffer_Release(&view);
        PyErr_SetString(PyExc_ValueError, "rows_to_lists: buffer smaller than Q * k int32");
        return NULL;
    }
    const int32_t *p = (const int32_t *)view.buf;
    PyObject *outer = PyList_New(Q);
    if (!outer) { PyBuffer_Release(&view); return NULL; }
    for (Py_ssize_t q = 0; q < Q; ++q) {
        PyObject *row = PyList_New(k);
        if (!row) { Py_DECREF(outer); PyBuffer_Release(&view); return NULL; }
        PyList_SET_ITEM(outer, q, row);   /* owned by outer from here: a failure below releases everything through it */
        for (Py_ssize_t j = 0; j < k; ++j) {
            PyObject *v = PyLong_FromLong((long)p[q * k + j]);
            if (!v) { Py_DECREF(outer); PyBuffer_Release(&view); return NULL; }
            PyList_SET_ITEM(row, j, v);
        }
        if (untrack) PyObject_GC_UnTrack(row);
    }
    PyBuffer_Release(&view);
    return outer;
}

static PyMethodDef methods[] = {{"rows_to_lists", rows_to_lists, METH_VARARGS, "int32 [Q, k] buffer -> list of Q lists of k ints"},
                                {NULL, NULL, 0, NULL}};
static struct PyModuleDef module = {PyModuleDef_HEAD_INIT, "_nlsh_fastlists", NULL, -1, methods, NULL, NULL, NULL, NULL};
PyMODINIT_FUNC PyInit__nlsh_fastlists(void) { return PyModule_Create(&module); }
