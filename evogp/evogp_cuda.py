"""``import evogp.evogp_cuda`` — in the reference this is the compiled extension whose import registers the five
``torch.ops.evogp_cuda.*`` operators (src/evogp/cuda/torch_wrapper.cu:287-307).  Here the registrars live in
``evogp_amd/lib/libevogp_torch.so``; importing this module loads it."""
import evogp_amd.ops  # noqa: F401
