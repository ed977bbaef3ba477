"""Decode-pool worker of mxio.ImageRecordIter: `python -m improving_face_recognition_performance_using_triplet_loss_amd.decode_worker`.

A plain child process speaking length-free pickle frames over its stdin / stdout: task = (path, [(offset, length)...], gray) ->
([label...], uint8 array (n, H, W[, C]) when every image of the slice has one size, else a list of arrays); None = exit.
Started with subprocess (not multiprocessing), so it never re-imports the caller's __main__ script and inherits nothing of the
parent's GPU state; it imports numpy + PIL only (ref: the decode threads inside MXNet's ImageRecordIter, train_efm.py:179-181)."""
import os
import pickle
import sys


def main():
    from improving_face_recognition_performance_using_triplet_loss_amd.mxio import _decode_slice
    inp, out = os.fdopen(os.dup(0), "rb"), os.fdopen(os.dup(1), "wb")
    sys.stdout = sys.stderr          # nothing but frames may reach the parent's pipe
    while True:
        try:
            task = pickle.load(inp)
        except EOFError:
            return
        if task is None:
            return
        try:
            res = _decode_slice(task)
        except BaseException as e:     # the parent re-raises
            res = e
        pickle.dump(res, out, protocol=pickle.HIGHEST_PROTOCOL)
        out.flush()


if __name__ == "__main__":
    main()
