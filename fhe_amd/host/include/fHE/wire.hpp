// fHE/wire.hpp — limb-dump wire format, version 1 ("no serialization exists in the reference ...
// define a simple limb-dump format first").  No counterpart in the reference; INTEGRATION.md holds the byte layout and
// fhe_amd/wire.py is the same format in Python.  One record = 40-byte little-endian header, n_polys * n_limbs * N
// uint64 limbs (limb-major, the layout the C-ABI takes), one uint64 checksum of the payload words.
#pragma once
#include "fHE/cipher.hpp"
#include <cstring>
#include <istream>
#include <ostream>
namespace fHE {
namespace wire {
enum Kind : uint16_t { POLY = 1, CIPHER = 2, KEY = 3 };
enum Flags : uint32_t { NTT_DOMAIN = 1u, HAS_MUL_AUX = 2u };
struct Header {
  char magic[4];  // "fHEw"
  uint16_t version, kind;
  uint32_t logN, n_limbs, n_polys, flags;
  double scale;
  uint64_t galois;
};
static_assert(sizeof(Header) == 40, "wire header is 40 bytes");

// position-weighted word sum mod 2^64: sum_k w[k] * ((k+1) * 0x9E3779B97F4A7C15 | 1)
inline uint64_t checksum(const uint64_t *w, size_t n, uint64_t start_index = 0, uint64_t acc = 0) {
  for (size_t k = 0; k < n; ++k) acc += w[k] * (((start_index + k + 1) * 0x9E3779B97F4A7C15ull) | 1ull);
  return acc;
}

inline void write_record(std::ostream &os, Kind kind, uint32_t flags, double scale, uint64_t galois,
                         std::vector<context::poly_t const *> const &polys) {
  if (polys.empty() || !polys[0]) throw std::invalid_argument("wire: nothing to write");
  Header h{};
  std::memcpy(h.magic, "fHEw", 4);
  h.version = 1; h.kind = kind; h.logN = (uint32_t)yell::runtime::get().logn;
  h.n_limbs = (uint32_t)polys[0]->moduli_count(); h.n_polys = (uint32_t)polys.size(); h.flags = flags;
  h.scale = scale; h.galois = galois;
  os.write(reinterpret_cast<const char *>(&h), sizeof h);
  const size_t words = (size_t)h.n_limbs * context::poly_t::degree();
  uint64_t sum = 0, at = 0;
  for (auto p : polys) {
    if (!p || p->moduli_count() != h.n_limbs) throw std::invalid_argument("wire: polynomials of one record must have equal limb counts");
    os.write(reinterpret_cast<const char *>(p->cptr_at(0)), words * 8);
    sum = checksum(p->cptr_at(0), words, at, sum); at += words;
  }
  os.write(reinterpret_cast<const char *>(&sum), 8);
  if (!os) throw std::runtime_error("wire: write failed");
}

// reads one record into `polys` (resized to the record's limb count); returns its header
inline Header read_record(std::istream &is, std::vector<context::poly_t> *polys) {
  Header h{};
  is.read(reinterpret_cast<char *>(&h), sizeof h);
  if (!is || std::memcmp(h.magic, "fHEw", 4)) throw std::runtime_error("wire: bad magic");
  if (h.version != 1) throw std::runtime_error("wire: unsupported version");
  if (h.logN != yell::runtime::get().logn) throw std::runtime_error("wire: ring degree differs from the active context");
  if (!h.n_polys || !h.n_limbs || h.n_limbs >= yell::params::kMaxNbModuli || h.n_polys > 4096) throw std::runtime_error("wire: bad shape");
  const size_t words = (size_t)h.n_limbs * context::poly_t::degree();
  polys->clear();
  uint64_t sum = 0, at = 0, want = 0;
  for (uint32_t i = 0; i < h.n_polys; ++i) {
    polys->emplace_back(h.n_limbs);
    is.read(reinterpret_cast<char *>(polys->back().ptr_at(0)), words * 8);
    if (!is) throw std::runtime_error("wire: truncated payload");
    sum = checksum(polys->back().cptr_at(0), words, at, sum); at += words;
  }
  is.read(reinterpret_cast<char *>(&want), 8);
  if (!is || want != sum) throw std::runtime_error("wire: checksum mismatch");
  return h;
}

inline void write_poly(std::ostream &os, context::poly_t const &p, bool ntt_domain) {
  write_record(os, POLY, ntt_domain ? NTT_DOMAIN : 0u, 1., 0, {&p});
}
inline bool read_poly(std::istream &is, context::poly_t *p, bool *ntt_domain = nullptr) {
  if (!p) return false;
  std::vector<context::poly_t> v;
  const Header h = read_record(is, &v);
  if (h.kind != POLY || v.size() != 1) throw std::runtime_error("wire: not a polynomial record");
  if (ntt_domain) *ntt_domain = h.flags & NTT_DOMAIN;
  *p = std::move(v[0]);
  return true;
}
// a ciphertext is (bx, ax[, mul_aux]) in the NTT domain plus its scale (cipher.hpp:5-50)
inline void write_cipher(std::ostream &os, Cipher const &c) {
  std::vector<context::poly_t const *> ps{&c.bx, &c.ax};
  if (c.mul_aux) ps.push_back(c.mul_aux.get());
  write_record(os, CIPHER, NTT_DOMAIN | (c.mul_aux ? HAS_MUL_AUX : 0u), c.scale(), 0, ps);
}
inline bool read_cipher(std::istream &is, Cipher *c) {
  if (!c) return false;
  std::vector<context::poly_t> v;
  const Header h = read_record(is, &v);
  const bool aux = h.flags & HAS_MUL_AUX;
  if (h.kind != CIPHER || v.size() != (aux ? 3u : 2u) || !(h.scale > 0.)) throw std::runtime_error("wire: not a ciphertext record");
  c->bx = std::move(v[0]); c->ax = std::move(v[1]);
  c->mul_aux = aux ? std::make_shared<context::poly_t>(std::move(v[2])) : nullptr;
  c->scale(h.scale);
  return true;
}
// key material: MultKey = 2*L polynomials (beta_0..beta_{L-1}, alpha_0..alpha_{L-1}), RotKey = (beta, alpha) + galois
inline void write_key(std::ostream &os, std::vector<context::poly_t const *> const &polys, uint64_t galois = 0) {
  write_record(os, KEY, NTT_DOMAIN, 1., galois, polys);
}
}  // namespace wire
}  // namespace fHE
