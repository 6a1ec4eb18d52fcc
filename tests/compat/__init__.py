"""Harness-side compatibility shim that lets the reference's UNCHANGED tools (tools/test.py, tools/demo.py) import and
run in this container (no cv2, NumPy 2 removed np.float/np.int/np.int0, utils/pyvotkit is an
unbuilt Cython extension).  Test infrastructure only: nothing under siammask_amd/ imports it, and it never edits or
copies the tools."""
