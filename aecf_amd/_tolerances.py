"""Per-tensor bounds of the bf16-STORED path against float32 math.  Kept inside the package so
that the runtime smoke check (``__graft_entry__.smoke``) and the tests assert the SAME numbers without the package depending
on the tests tree.  Not used by any product path."""

# ---- the bf16 contract, per tensor: measured on MI355X over the g2 / hot fixtures, the shape sweep and
# the random cases (round 3, gpurun_out/parity_errors.jsonl: maxima in the comments) + ~10-20 % margin.  A bf16-STORED tensor
# carries one output rounding (up to 2^-9 of ITS OWN magnitude, 3.9e-3 of the largest element in the worst case); the
# parameter gradients are float32 batch sums of products whose operands (do = dy W_o, the pooled rows) were rounded to bf16
# once each.  Nothing here is a blanket tolerance: a tensor that drifts by 20 % fails.
#   bf16-stored gradients (bf16 parameters):           y 3.25e-3  wbar 3.34e-3  dx 4.09e-3 (smoke's scaled modalities)  dquery 4.04e-3  dw_in 4.36e-3
#                                                      db_in 4.31e-3  dw_out 3.92e-3  db_out 3.02e-3
BF16_BOUNDS = dict(y=4.0e-3, wbar=4.0e-3, dx=4.5e-3, dquery=5.0e-3, dw_in=5.2e-3, db_in=5.2e-3, dw_out=4.7e-3, db_out=3.7e-3)
#   float32-stored gradients of the bf16 kernels (float32 master parameters; y / wbar / dx still bf16-stored):
#                                                      y 3.79e-3  wbar 3.76e-3  dx 4.10e-3  dq 4.41e-3  dw_in 3.97e-3
#                                                      db_in 3.63e-3  dw_out 2.46e-3  db_out 8e-8 (a float32 column sum of dy)
BF16_F32GRAD_BOUNDS = dict(y=4.2e-3, wbar=4.2e-3, dx=4.5e-3, dq=5.0e-3, dquery=5.0e-3, dw_in=4.4e-3, db_in=4.0e-3, dw_out=3.0e-3,
                           db_out=1e-5)
#   ... where the hi + lo weight-gradient products are built (bf16, d = 256 / 512, M <= 3: on by themselves for float32-stored
#   gradients, layer.PoolOptions.hilo_grads; round 5) the three gradients that are sums of products of DERIVED operands meet
#   north_star's 1e-3 with an order of magnitude to spare: measured 3-5e-6 at the headline shape.  dq / dquery here is the
#   gradient handed back through a bf16 query tensor (one output rounding) and keeps its bound.
BF16_F32GRAD_HILO_BOUNDS = dict(BF16_F32GRAD_BOUNDS, dw_in=1e-4, db_in=1e-4, dw_out=1e-4)
