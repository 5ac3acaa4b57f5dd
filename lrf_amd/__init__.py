"""lrf_amd — MI355X-native QMF image-compression hot path, drop-in for pashtari/lrf's
`lrf.qmf_encode` / `lrf.qmf_decode` / `lrf.QMF`."""
from .codec import (qmf_decode, qmf_decode_batch, qmf_encode, qmf_encode_batch, qmf_factorize_batch, qmf_factorize_host,
                    qmf_ranks)
from .container import (bytes_to_dict, combine_bytes, decode_matrix, decode_tensor, dict_to_bytes, encode_matrix,
                        encode_tensor, separate_bytes)
from .factorization import QMF
from .harness import eval_compression, rd_sweep
from .rd import LOESS, interpolate_records
from .svd_codec import svd_decode, svd_encode
from .metrics import bits_per_pixel, compression_ratio, mse, psnr, ssim

__all__ = ["qmf_encode", "qmf_decode", "qmf_encode_batch", "qmf_decode_batch", "qmf_factorize_batch", "qmf_factorize_host", "qmf_ranks", "svd_encode", "svd_decode",
           "QMF", "eval_compression", "rd_sweep", "LOESS", "interpolate_records", "psnr", "ssim", "mse", "bits_per_pixel", "compression_ratio", "combine_bytes", "separate_bytes",
           "dict_to_bytes", "bytes_to_dict", "encode_matrix", "decode_matrix", "encode_tensor", "decode_tensor"]
