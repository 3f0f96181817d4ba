// host_roundtrip — BASELINE config 1 through the C++ host mirror: keygen -> encrypt -> multiply_and_relin
// (one fused GPU call) -> decrypt, plus a rotation.  Mirrors what the reference's main.cpp does with its own API.
// Writes every key, input and output limb to a dump file so that tests/ can re-run the same inputs through the CPU
// checker and compare bit for bit.   usage: host_roundtrip <logN> <L> <K> <dump-file>
#include <cstdio>
#include <chrono>
#include <cstdlib>
#include <fstream>
#include <new>
#include "fHE/context.hpp"
#include "fHE/cipher.hpp"
#include "fHE/decryptor.hpp"
#include "fHE/encoder.hpp"
#include "fHE/encryptor.hpp"
#include "fHE/evaluator.hpp"
#include "fHE/wire.hpp"

using T = yell::params::value_type;
static void dump(FILE *f, const char *name, fHE::context::poly_t const &p) {
  uint64_t n = p.moduli_count();
  char tag[16] = {0};
  std::snprintf(tag, sizeof(tag), "%s", name);
  std::fwrite(tag, 1, 16, f);
  std::fwrite(&n, 8, 1, f);
  std::fwrite(p.cptr_at(0), 8, n * fHE::context::degree, f);
}

// host_roundtrip bench <logN> <L> <K> <batch> <reps>: multiply_and_relin per second THROUGH THE C++ API with the ciphertexts
// resident on the GPU (fHE::Cipher keeps its limbs in device memory; yell/poly.hpp), batched overload and one-at-a-time.
// Each repetition starts from fresh Li-limb copies (device-to-device, outside the timed region); the timed region is the
// Evaluator call(s) up to the completion of the device work.
static int bench_mode(int argc, char **argv) {
  const uint32_t logN = argc > 2 ? std::atoi(argv[2]) : 16, L = argc > 3 ? std::atoi(argv[3]) : 30, K = argc > 4 ? std::atoi(argv[4]) : 1;
  const size_t B = argc > 5 ? std::atoi(argv[5]) : 16, reps = argc > 6 ? std::atoi(argv[6]) : 5;
  fHE::Context context(logN, L, K, 0);
  fHE::SK sk;
  fHE::PK pk(sk);
  fHE::MultKey mkey(sk);
  fHE::Encryptor encryptor;
  fHE::Evaluator evaluator;
  fHE::context::poly_t msg(L);
  fHE::Cipher c0, c1;
  if (!encryptor.encrypt(&c0, msg, pk) || !encryptor.encrypt(&c1, msg, pk)) return 40;
  std::vector<fHE::Cipher> src0(B, c0), src1(B, c1);
  std::vector<fHE::Cipher const *> rhs;
  for (auto &c : src1) rhs.push_back(&c);
  { fHE::Cipher warm(c0); if (!evaluator.multiply_and_relin(&warm, c1, mkey)) return 41; }  // uploads the key
  auto timed = [&](bool batched) {
    double total = 0;
    for (size_t r = 0; r < reps + 1; ++r) {
      std::vector<fHE::Cipher> work(src0);
      std::vector<fHE::Cipher *> lhs;
      for (auto &c : work) lhs.push_back(&c);
      fhe_hip_sync(context.handle());
      const auto t0 = std::chrono::steady_clock::now();
      bool ok = true;
      if (batched) ok = evaluator.multiply_and_relin(lhs, rhs, mkey);
      else for (size_t b = 0; b < B && ok; ++b) ok = evaluator.multiply_and_relin(lhs[b], *rhs[b], mkey);
      fhe_hip_sync(context.handle());
      const auto t1 = std::chrono::steady_clock::now();
      if (!ok) return -1.0;
      if (r) total += std::chrono::duration<double>(t1 - t0).count();  // repetition 0 is the warm-up (workspaces, gather)
    }
    return (double)(B * reps) / total;
  };
  const double rb = timed(true), r1 = timed(false);
  if (rb < 0 || r1 < 0) return 42;
  std::printf("{\"api\": \"fHE::Evaluator::multiply_and_relin (C++ host mirror, ciphertexts device-resident)\", \"logN\": %u, \"Li\": %u, \"batch\": %zu, "
              "\"reps\": %zu, \"batched_ops_per_s\": %.1f, \"one_at_a_time_ops_per_s\": %.1f}\n", logN, L, B, reps, rb, r1);
  return 0;
}

int main(int argc, char **argv) {
  if (argc > 1 && std::string(argv[1]) == "bench") {
    try { return bench_mode(argc, argv); } catch (std::exception const &e) { std::fprintf(stderr, "exception: %s\n", e.what()); return 1; }
  }
  const uint32_t logN = argc > 1 ? std::atoi(argv[1]) : 13, L = argc > 2 ? std::atoi(argv[2]) : 4, K = argc > 3 ? std::atoi(argv[3]) : 4;
  const char *path = argc > 4 ? argv[4] : "host_roundtrip.bin";
  try {
    fHE::Context context(logN, L, K, 0);
    const size_t N = fHE::context::degree;
    fHE::SK sk;
    fHE::PK pk(sk);
    fHE::MultKey mkey(sk);
    fHE::RotKey rkey(sk, 1, fHE::RotKey::Direction::LEFT);
    auto encoder = std::make_shared<fHE::Encoder>();
    fHE::Encryptor encryptor(encoder);
    fHE::Decryptor decryptor(encoder);
    fHE::Evaluator evaluator(encoder);

    // integer plaintexts a, b with small coefficients times 2^40
    fHE::context::poly_t ma(L), mb(L);
    std::vector<long> a(N), b(N);
    yell::prng rng(42);
    for (size_t d = 0; d < N; ++d) { a[d] = (long)(rng.next() % 9) - 4; b[d] = (long)(rng.next() % 9) - 4; }
    for (size_t cm = 0; cm < L; ++cm) {
      const T p = yell::params::P[cm];
      for (size_t d = 0; d < N; ++d) {
        const T sa = (T)std::labs(a[d]) << 40, sb = (T)std::labs(b[d]) << 40;
        ma.ptr_at(cm)[d] = a[d] >= 0 ? sa : p - sa;
        mb.ptr_at(cm)[d] = b[d] >= 0 ? sb : p - sb;
      }
    }
    fHE::Cipher ca, cb;
    encryptor.encrypt(&ca, ma, pk);
    encryptor.encrypt(&cb, mb, pk);
    ca.scale(1099511627776.0); cb.scale(1099511627776.0);
    fHE::Cipher in_a(ca), in_b(cb);

    // fresh decrypt: message + small noise
    fHE::context::poly_t dec(L);
    decryptor.decrypt(&dec, ca, sk);
    long worst = 0;
    for (size_t d = 0; d < N; ++d) {
      const T p = yell::params::P[0]; T v = dec.cptr_at(0)[d];
      long c = v > p / 2 ? -(long)(p - v) : (long)v;
      worst = std::max(worst, std::labs(c - (a[d] << 40)));
    }
    std::printf("fresh decrypt noise (limb 0): %ld\n", worst);
    if (worst > (1L << 20)) { std::fprintf(stderr, "decrypt noise too large\n"); return 2; }

    if (!evaluator.multiply_and_relin(&ca, cb, mkey)) { std::fprintf(stderr, "multiply_and_relin failed\n"); return 3; }
    if (ca.moduli_count() != L - 1 || !ca.canonical()) { std::fprintf(stderr, "unexpected result shape\n"); return 4; }
    fHE::context::poly_t dec_mul(L - 1);
    decryptor.decrypt(&dec_mul, ca, sk);

    // separate multiply + relinearize must give the same ciphertext (evaluator.cpp:119-126)
    fHE::Cipher two_step(in_a);
    if (!evaluator.multiply(&two_step, in_b) || two_step.canonical() || !evaluator.relinearize(&two_step, mkey)) { std::fprintf(stderr, "two-step path failed\n"); return 5; }
    if (std::memcmp(two_step.bx.cptr_at(0), ca.bx.cptr_at(0), (L - 1) * N * 8) || std::memcmp(two_step.ax.cptr_at(0), ca.ax.cptr_at(0), (L - 1) * N * 8)) {
      std::fprintf(stderr, "multiply+relinearize differs from multiply_and_relin\n"); return 6;
    }
    // misuse conventions: relinearize on a canonical ciphertext returns false; rotate of a non-canonical one too
    if (evaluator.relinearize(&ca, mkey)) { std::fprintf(stderr, "relinearize accepted a canonical ciphertext\n"); return 7; }

    // two Evaluators sharing the one device context, two MultKeys used alternately, and a key rebuilt where another one
    // was destroyed: the device must always relinearise with the key that was passed
    {
      fHE::Evaluator other(encoder);
      fHE::MultKey mkey2(sk);
      fHE::Cipher r1(in_a), r2(in_a), r3(in_a), r4(in_a);
      if (!evaluator.multiply_and_relin(&r1, in_b, mkey) || !other.multiply_and_relin(&r2, in_b, mkey2) ||
          !evaluator.multiply_and_relin(&r3, in_b, mkey) || !other.multiply_and_relin(&r4, in_b, mkey2)) { std::fprintf(stderr, "alternating keys failed\n"); return 28; }
      const size_t bytes = (L - 1) * N * 8;
      if (std::memcmp(r1.bx.cptr_at(0), ca.bx.cptr_at(0), bytes) || std::memcmp(r3.bx.cptr_at(0), ca.bx.cptr_at(0), bytes) ||
          std::memcmp(r2.bx.cptr_at(0), r4.bx.cptr_at(0), bytes) || !std::memcmp(r1.bx.cptr_at(0), r2.bx.cptr_at(0), bytes)) {
        std::fprintf(stderr, "a stale MultKey was used on the device\n"); return 29;
      }
      fHE::context::poly_t d2(L - 1);
      decryptor.decrypt(&d2, r2, sk);  // the second key is a valid key too: same plaintext up to noise
      long off = 0;
      for (size_t d = 0; d < N; ++d) {
        const T p = yell::params::P[0];
        const T x = d2.cptr_at(0)[d], y = dec_mul.cptr_at(0)[d], df = x >= y ? x - y : y - x;
        off = std::max(off, (long)std::min(df, p - df));
      }
      if (off > (1L << 40)) { std::fprintf(stderr, "second MultKey decrypts differently (%ld)\n", off); return 30; }
      alignas(fHE::RotKey) static unsigned char slot[sizeof(fHE::RotKey)];
      fHE::Cipher q1(in_a), q2(in_a);
      auto *k1 = new (slot) fHE::RotKey(sk, 1, fHE::RotKey::Direction::LEFT);
      const bool ok1 = evaluator.rotate_slots(&q1, *k1);
      k1->~RotKey();
      auto *k2 = new (slot) fHE::RotKey(sk, 1, fHE::RotKey::Direction::LEFT);  // same address, same galois, different key
      const bool ok2 = other.rotate_slots(&q2, *k2);
      fHE::Cipher q3(in_a);
      fHE::context::poly_t want_b(q2.bx);
      const bool ok3 = evaluator.rotate_slots(&q3, *k2);
      k2->~RotKey();
      if (!ok1 || !ok2 || !ok3 || !std::memcmp(q1.bx.cptr_at(0), q2.bx.cptr_at(0), L * N * 8) || std::memcmp(q3.bx.cptr_at(0), want_b.cptr_at(0), L * N * 8)) {
        std::fprintf(stderr, "a RotKey rebuilt in place was mistaken for the old one\n"); return 31;
      }
      std::printf("key identity: alternating MultKeys and a RotKey rebuilt in place are told apart\n");
    }

    fHE::Cipher rot(in_a);
    if (!evaluator.rotate_slots(&rot, rkey)) { std::fprintf(stderr, "rotate_slots failed\n"); return 8; }
    fHE::context::poly_t dec_rot(L);
    decryptor.decrypt(&dec_rot, rot, sk);

    // MixedRotKey variant, and an arbitrary offset through a RotKeySet (3 = 1 + 2)
    fHE::MixedRotKey mkey_rot(sk, 1, fHE::MixedRotKey::Direction::LEFT);
    fHE::Cipher rotm(in_a);
    if (!evaluator.rotate_slots(&rotm, mkey_rot)) { std::fprintf(stderr, "rotate_slots(MixedRotKey) failed\n"); return 10; }
    fHE::context::poly_t dec_rotm(L);
    decryptor.decrypt(&dec_rotm, rotm, sk);
    fHE::RotKeySet<fHE::RotKey> rset(sk);
    fHE::Cipher rot3(in_a);
    if (!evaluator.rotate_slots(&rot3, 3, rset)) { std::fprintf(stderr, "rotate_slots(offset) failed\n"); return 11; }
    fHE::context::poly_t dec_rot3(L);
    decryptor.decrypt(&dec_rot3, rot3, sk);

    // Encoder (reference main.cpp:27-31 style values 1.0 + i) and the power-basis rotation / replicate composites
    const size_t nr_slots = N >> 1;
    std::vector<double> vals(nr_slots);
    for (size_t i = 0; i < nr_slots; ++i) vals[i] = 1.0 + (double)(i % 97) - 0.25 * (double)(i % 7);
    fHE::context::poly_t enc_poly(L);
    const auto tc0 = std::chrono::steady_clock::now();
    encoder->encode(&enc_poly, vals, 1099511627776.0);
    const auto tc1 = std::chrono::steady_clock::now();
    std::vector<double> back;
    encoder->decode(&back, enc_poly, 1099511627776.0);
    const auto tc2 = std::chrono::steady_clock::now();
    std::printf("encoder (host, 1 thread): encode %.2f ms, decode %.2f ms at N=2^%u, %u limbs\n",
                std::chrono::duration<double, std::milli>(tc1 - tc0).count(), std::chrono::duration<double, std::milli>(tc2 - tc1).count(), logN, L);
    double enc_err = 0;
    for (size_t i = 0; i < nr_slots; ++i) enc_err = std::max(enc_err, std::fabs(back[i] - vals[i]));
    std::printf("encode/decode max error: %.3e\n", enc_err);
    if (!(enc_err < 1e-6)) { std::fprintf(stderr, "encoder round trip too inaccurate\n"); return 12; }
    {
      // the device encoder against the independent host implementation (twist + cyclic DFT, multi-word CRT): coefficients agree to
      // the rounding of the two FFTs (<= 2^-40 of the scaled magnitude), decoded slots to 1e-9
      fHE::context::poly_t enc_host(L);
      encoder->encode_on_host(&enc_host, std::vector<std::complex<double>>(vals.begin(), vals.end()), 1099511627776.0);
      long worst_c = 0;
      for (size_t cm = 0; cm < L; ++cm) {
        const T p = yell::params::P[cm];
        for (size_t d = 0; d < N; ++d) {
          const T x = enc_poly.cptr_at(cm)[d], y = enc_host.cptr_at(cm)[d], df = x >= y ? x - y : y - x;
          worst_c = std::max(worst_c, (long)std::min(df, p - df));
        }
      }
      std::vector<std::complex<double>> back_host, back_dev;
      encoder->decode_on_host(&back_host, enc_poly, 1099511627776.0);
      encoder->decode(&back_dev, enc_poly, 1099511627776.0);
      double worst_s = 0;
      for (size_t i = 0; i < nr_slots; ++i) worst_s = std::max(worst_s, std::abs(back_host[i] - back_dev[i]));
      std::printf("device encoder vs host cross-check: coefficients differ by at most %ld, decoded slots by %.3e\n", worst_c, worst_s);
      if (worst_c > 256 || !(worst_s < 1e-9)) { std::fprintf(stderr, "device and host encoders disagree\n"); return 32; }
    }
    fHE::Cipher nn_in(in_a);
    nn_in.bx.backward(); nn_in.ax.backward();
    fHE::Cipher nn(nn_in);
    if (!evaluator.rotate_slots_non_ntt(&nn, rkey)) { std::fprintf(stderr, "rotate_slots_non_ntt failed\n"); return 13; }
    fHE::Cipher nn5(nn_in);
    if (!evaluator.rotate_slots_non_ntt(&nn5, -5, rset)) { std::fprintf(stderr, "rotate_slots_non_ntt(offset) failed\n"); return 14; }
    const size_t rep_pos = 5;
    fHE::Cipher rep(in_a);
    if (!evaluator.replicate(&rep, rep_pos, rset)) { std::fprintf(stderr, "replicate failed\n"); return 15; }
    fHE::context::poly_t rep_mask(L);
    { std::vector<double> mask(nr_slots, 0.); mask[rep_pos] = 1.; encoder->encode(&rep_mask, mask, fHE::context::encoder_scale / 8.); }

    // CKKS end to end through the Encoder overloads (reference main.cpp flow): slot values in, slot values out.
    // decode(decrypt(x (*) y)) ~ x[i]*y[i]; decode(decrypt(rotate_left_1(x))) ~ x[i+1]
    {
      std::vector<double> xs(nr_slots), ys(nr_slots), got;
      for (size_t i = 0; i < nr_slots; ++i) { xs[i] = std::sin(0.37 * (double)i) * 1.5; ys[i] = std::cos(0.11 * (double)i) - 0.5; }
      fHE::Cipher cx, cy;
      if (!encryptor.encrypt(&cx, xs, pk) || !encryptor.encrypt(&cy, ys, pk)) { std::fprintf(stderr, "encrypt(values) failed\n"); return 16; }
      fHE::Cipher cr(cx);
      if (!evaluator.multiply_and_relin(&cx, cy, mkey) || !decryptor.decrypt(&got, cx, sk)) { std::fprintf(stderr, "value multiply failed\n"); return 17; }
      double mul_err = 0;
      for (size_t i = 0; i < nr_slots; ++i) mul_err = std::max(mul_err, std::fabs(got[i] - xs[i] * ys[i]));
      std::printf("slot-wise product max error: %.3e (scale after rescale 2^%.2f)\n", mul_err, std::log2(cx.scale()));
      if (!(mul_err < 1e-6)) { std::fprintf(stderr, "slot-wise product too inaccurate\n"); return 18; }
      if (K >= L) {  // hybrid key switching is only low-noise when P >= Q (the reference fixes K = L)
        if (!evaluator.rotate_slots(&cr, rkey) || !decryptor.decrypt(&got, cr, sk)) { std::fprintf(stderr, "value rotate failed\n"); return 19; }
        double rot_err = 0;
        for (size_t i = 0; i < nr_slots; ++i) rot_err = std::max(rot_err, std::fabs(got[i] - xs[(i + 1) % nr_slots]));
        std::printf("slot rotation max error: %.3e\n", rot_err);
        if (!(rot_err < 1e-6)) { std::fprintf(stderr, "slot rotation too inaccurate\n"); return 20; }
        double one = 0;
        if (!decryptor.decrypt(&one, cr, sk) || std::fabs(one - xs[1]) > 1e-6) { std::fprintf(stderr, "decrypt(double) wrong\n"); return 21; }
      }
    }

    // BaseConverter mirror (base_converter.hpp:18-35): mod_up, the two single-target conversions, mod_down, rescale
    fHE::BaseConverter bconv;
    fHE::context::poly_t bc_in(in_a.ax);
    bc_in.backward();  // power basis
    fHE::context::poly_t bc_up(L + K), bc_down(L), bc_resc(L - 1), bc_sp(K), bc_one(2);
    if (!bconv.approximated_mod_up(&bc_up, bc_in)) { std::fprintf(stderr, "approximated_mod_up failed\n"); return 23; }
    std::vector<T> single(N);
    bconv.approx_convert_to_special_basis(single.data(), bc_in, K - 1);
    if (std::memcmp(single.data(), bc_up.cptr_at(L + K - 1), N * 8)) { std::fprintf(stderr, "approx_convert_to_special_basis differs from mod_up\n"); return 24; }
    std::memcpy(bc_sp.ptr_at(0), bc_up.cptr_at(L), K * N * 8);
    std::memcpy(bc_one.ptr_at(0), single.data(), N * 8);  // reuse as scratch for the second single-target call
    bconv.neg_approx_convert_to_normal_basis(bc_one.ptr_at(1), bc_sp, L - 1);
    fHE::context::poly_t bc_mixed(bc_up);  // normal limbs to the NTT basis, special limbs stay in the power basis (:205-209)
    for (size_t j = 0; j < L; ++j) yell::ntt::forward(bc_mixed.ptr_at(j), j);
    if (!bconv.approximated_mod_down(&bc_down, bc_mixed)) { std::fprintf(stderr, "approximated_mod_down failed\n"); return 25; }
    fHE::context::poly_t bc_r_in(in_a.ax);
    yell::ntt::backward(bc_r_in.ptr_at(L - 1), L - 1);  // last limb in the power basis (:246)
    if (!bconv.rescale(&bc_resc, bc_r_in)) { std::fprintf(stderr, "rescale failed\n"); return 26; }
    try { fHE::context::poly_t one(1), ext(1 + K); bconv.approximated_mod_up(&ext, one); std::fprintf(stderr, "mod_up accepted one limb\n"); return 27; }
    catch (std::invalid_argument const &) {}

    FILE *f = std::fopen(path, "wb");
    if (!f) { std::perror("fopen"); return 9; }
    uint64_t hdr[4] = {logN, L, K, rkey.galois};
    std::fwrite(hdr, 8, 4, f);
    dump(f, "sx", sk.sx); dump(f, "ext_sx", sk.ext_sx);
    for (size_t j = 0; j < L; ++j) { dump(f, "mk_beta", *mkey.get_beta_at(j)); dump(f, "mk_alpha", *mkey.get_alpha_at(j)); }
    dump(f, "rk_beta", rkey.beta); dump(f, "rk_alpha", rkey.alpha);
    dump(f, "a_bx", in_a.bx); dump(f, "a_ax", in_a.ax); dump(f, "b_bx", in_b.bx); dump(f, "b_ax", in_b.ax);
    dump(f, "mul_bx", ca.bx); dump(f, "mul_ax", ca.ax); dump(f, "dec_mul", dec_mul);
    dump(f, "rot_bx", rot.bx); dump(f, "rot_ax", rot.ax); dump(f, "dec_rot", dec_rot);
    dump(f, "msg_a", ma); dump(f, "msg_b", mb);
    for (size_t j = 0; j < L; ++j) { dump(f, "xk_beta", *mkey_rot.get_beta_at(j)); dump(f, "xk_alpha", *mkey_rot.get_alpha_at(j)); }
    dump(f, "rotm_bx", rotm.bx); dump(f, "rotm_ax", rotm.ax); dump(f, "dec_rotm", dec_rotm); dump(f, "dec_rot3", dec_rot3);
    dump(f, "enc_poly", enc_poly);
    dump(f, "nn_in_bx", nn_in.bx); dump(f, "nn_in_ax", nn_in.ax); dump(f, "nn_bx", nn.bx); dump(f, "nn_ax", nn.ax);
    dump(f, "nn5_bx", nn5.bx); dump(f, "nn5_ax", nn5.ax);
    for (size_t i = 0; i + 1 < logN; ++i) { auto k = rset.get((uint16_t)(1u << i)); dump(f, "rs_beta", k->beta); dump(f, "rs_alpha", k->alpha); }
    dump(f, "rep_mask", rep_mask); dump(f, "rep_bx", rep.bx); dump(f, "rep_ax", rep.ax);
    dump(f, "bc_in", bc_in); dump(f, "bc_up", bc_up); dump(f, "bc_mixed", bc_mixed); dump(f, "bc_down", bc_down);
    dump(f, "bc_r_in", bc_r_in); dump(f, "bc_resc", bc_resc); dump(f, "bc_one", bc_one);
    std::fclose(f);
    // the multiplied ciphertext and the rotation key once more in the limb-dump wire format (fHE/wire.hpp), read back here
    {
      const std::string wpath = std::string(path) + ".wire";
      { std::ofstream os(wpath, std::ios::binary); fHE::wire::write_cipher(os, ca); fHE::wire::write_key(os, {&rkey.beta, &rkey.alpha}, rkey.galois); }
      std::ifstream is(wpath, std::ios::binary);
      fHE::Cipher back_ct(1);
      fHE::wire::read_cipher(is, &back_ct);
      if (back_ct.moduli_count() != ca.moduli_count() || back_ct.scale() != ca.scale() ||
          std::memcmp(back_ct.bx.cptr_at(0), ca.bx.cptr_at(0), ca.moduli_count() * N * 8) ||
          std::memcmp(back_ct.ax.cptr_at(0), ca.ax.cptr_at(0), ca.moduli_count() * N * 8)) { std::fprintf(stderr, "wire round trip differs\n"); return 22; }
    }
    std::printf("host_roundtrip OK\n");
    return 0;
  } catch (std::exception const &e) {
    std::fprintf(stderr, "exception: %s\n", e.what());
    return 1;
  }
}
