// wire_selftest — CPU only (no device context): reads one ciphertext record in the limb-dump wire format, checks it,
// and writes it back followed by a polynomial record holding bx.  tests/test_wire.py drives it against fhe_amd/wire.py.
// usage: wire_selftest <logN> <in-file> <out-file>      exit 0 = OK, 2 = record rejected (message on stderr)
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include "fHE/wire.hpp"

int main(int argc, char **argv) {
  if (argc < 4) { std::fprintf(stderr, "usage: wire_selftest <logN> <in> <out>\n"); return 1; }
  auto &rt = yell::runtime::get();
  rt.logn = (size_t)std::atoi(argv[1]); rt.degree = (size_t)1 << rt.logn;
  fHE::context::degree = rt.degree;
  try {
    std::ifstream in(argv[2], std::ios::binary);
    fHE::Cipher c(1);
    fHE::wire::read_cipher(in, &c);
    std::printf("limbs=%zu canonical=%d scale=%.17g\n", c.moduli_count(), (int)c.canonical(), c.scale());
    std::ofstream out(argv[3], std::ios::binary);
    fHE::wire::write_cipher(out, c);
    fHE::wire::write_poly(out, c.bx, true);
    return 0;
  } catch (std::exception const &e) {
    std::fprintf(stderr, "%s\n", e.what());
    return 2;
  }
}
