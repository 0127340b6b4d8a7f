"""The half stem's input values.  Ultralytics (engine/predictor.py:preprocess) does im.half() and then im /= 255 in fp16: the exact
quotient rounded once.  The general half stem rounds the fp32 stem's table (float)i / 255.f to fp16, and the k 3 / stride 2 kernel
(misc_kernels.hip:stem3s2_u8_h) multiplies by 1/255.f and rounds.  The product differs from the table in fp32 for 126 byte values;
after the rounding to fp16 all three give the same 256 values, and this test keeps that true."""
import numpy as np
import torch


def test_table_product_and_half_division_give_the_same_256_halfs():
    b = np.arange(256, dtype=np.float32)
    table = (b / np.float32(255.0)).astype(np.float32)
    prod = (b * (np.float32(1.0) / np.float32(255.0))).astype(np.float32)
    assert (table != prod).any()                       # the fp32 stem could not use the product
    np.testing.assert_array_equal(table.astype(np.float16), prod.astype(np.float16))
    ultralytics = (torch.arange(256, dtype=torch.uint8).half() / 255).numpy()
    np.testing.assert_array_equal(table.astype(np.float16), ultralytics)
