"""tiny-ram-halo2 MI355X prover backend: HIP kernels + C ABI (csrc/, libtrh.so) and the
host-side mirror of halo2_proofs' arithmetic / poly::commitment interface for the MSM, NTT
and IPA-commitment hot path."""
__version__ = "0.1.0"
