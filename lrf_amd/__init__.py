"""lrf_amd — MI355X-native QMF image-compression hot path, drop-in for pashtari/lrf's
`lrf.qmf_encode` / `lrf.qmf_decode` / `lrf.QMF`."""
import os as _os

# The pipelined encoder (lrf_pipe) keeps an upload stream and two kernel streams busy next to the caller's own streams.
# The HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that share a queue
# serialise: measured on MI355X, 256 x 512x768 host->host takes 8.7 ms with 4 queues and 6.4 ms with 8.  The variable is
# read when the runtime initialises, so this only helps if lrf_amd is imported before the first GPU call; a value the
# user has set is left alone.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from .codec import (qmf_decode, qmf_decode_batch, qmf_encode, qmf_encode_batch, qmf_encode_sweep, qmf_factorize_batch,
                    qmf_factorize_host, qmf_ranks)
from .container import (bytes_to_dict, combine_bytes, decode_matrix, decode_tensor, dict_to_bytes, encode_matrix,
                        encode_tensor, separate_bytes)
from .factorization import QMF
from .harness import eval_compression, rd_sweep, rd_sweep_batched
from .rd import LOESS, interpolate_records
from .svd_codec import svd_decode, svd_encode
from .metrics import bits_per_pixel, compression_ratio, mse, psnr, ssim

__all__ = ["qmf_encode", "qmf_decode", "qmf_encode_batch", "qmf_encode_sweep", "qmf_decode_batch", "qmf_factorize_batch", "qmf_factorize_host", "qmf_ranks", "svd_encode", "svd_decode",
           "QMF", "eval_compression", "rd_sweep", "rd_sweep_batched", "LOESS", "interpolate_records", "psnr", "ssim", "mse", "bits_per_pixel", "compression_ratio", "combine_bytes", "separate_bytes",
           "dict_to_bytes", "bytes_to_dict", "encode_matrix", "decode_matrix", "encode_tensor", "decode_tensor"]
