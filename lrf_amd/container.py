"""Byte container of the reference codec, restated (host-side, byte-exact).

Layout (lrf/compression/utils.py:246-455):  combine([a, b, c, ...]) is a left fold of
len32_be(p1) || p1 || p2;  a 2-D integer factor is stored column by column, each column
zlib-compressed at level 9, behind a JSON header {"num_fibers", "mode", "dtype"}.
"""
import functools
import json
import zlib
from typing import Sequence

import numpy as np


def _join2(first: bytes, second: bytes) -> bytes:
    # lrf/compression/utils.py:246-265
    if not isinstance(first, bytes) or not isinstance(second, bytes):
        raise TypeError("Both payload1 and payload2 must be bytes objects.")
    if len(first) > 0xFFFFFFFF:
        raise ValueError("payload1 is too large to encode.")
    return len(first).to_bytes(4, byteorder="big") + first + second


def _split2(blob: bytes):
    # lrf/compression/utils.py:268-287
    if not isinstance(blob, bytes):
        raise TypeError("Combined must be a bytes object.")
    if len(blob) < 4:
        raise ValueError("Combined data is too short to decode.")
    n = int.from_bytes(blob[:4], byteorder="big")
    return blob[4:4 + n], blob[4 + n:]


def combine_bytes(payloads: Sequence[bytes]) -> bytes:
    # lrf/compression/utils.py:290-300
    return functools.reduce(_join2, payloads)


def separate_bytes(combined: bytes, num_payloads: int = 2):
    # lrf/compression/utils.py:303-321: peel payloads off the END of the left fold
    parts = []
    head = combined
    for _ in range(num_payloads - 1):
        head, tail = _split2(head)
        parts.insert(0, tail)
    parts.insert(0, head)
    return tuple(parts)


def dict_to_bytes(dictionary: dict) -> bytes:
    return json.dumps(dictionary).encode("utf-8")  # lrf/compression/utils.py:324-336


def bytes_to_dict(encoded: bytes) -> dict:
    return json.loads(encoded.decode("utf-8"))  # lrf/compression/utils.py:339-351


def encode_matrix(matrix: np.ndarray, mode: str = "col") -> bytes:
    """lrf/compression/utils.py:354-390 for a numpy 2-D array."""
    assert matrix.ndim == 2, "'matrix' must be a 2D tensor."
    assert mode in {"col", "row"}, "'mode' must be either 'col' or 'row'."
    fibers = [matrix[:, j:j + 1] for j in range(matrix.shape[1])] if mode == "col" else \
             [matrix[i:i + 1, :] for i in range(matrix.shape[0])]
    packed = [zlib.compress(np.ascontiguousarray(f).tobytes(), level=9) for f in fibers]
    header = dict_to_bytes({"num_fibers": len(fibers), "mode": mode, "dtype": str(matrix.dtype)})
    return combine_bytes([header, combine_bytes(packed)])


def decode_matrix(blob: bytes) -> np.ndarray:
    """lrf/compression/utils.py:393-426."""
    header, body = separate_bytes(blob)
    meta = bytes_to_dict(header)
    fibers = [np.frombuffer(zlib.decompress(f), dtype=np.dtype(meta["dtype"]))
              for f in separate_bytes(body, num_payloads=meta["num_fibers"])]
    return np.stack(fibers, axis=1 if meta["mode"] == "col" else 0)


def encode_tensor(array: np.ndarray) -> bytes:
    """lrf/compression/utils.py:429-455."""
    if array.ndim == 2:
        return encode_matrix(array)
    body = zlib.compress(np.ascontiguousarray(array).tobytes(), level=9)
    header = dict_to_bytes({"shape": list(array.shape), "dtype": str(array.dtype)})
    return combine_bytes([header, body])


def decode_tensor(blob: bytes) -> np.ndarray:
    """lrf/compression/utils.py:458-490."""
    header, body = separate_bytes(blob)
    meta = bytes_to_dict(header)
    if "num_fibers" in meta:
        return decode_matrix(blob)
    return np.frombuffer(zlib.decompress(body), dtype=np.dtype(meta["dtype"])).reshape(meta["shape"])
