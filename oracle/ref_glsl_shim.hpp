// TEST INFRASTRUCTURE — not product code.
//
// A small GLSL-dialect shim so that the reference's A_GPU/A_GLSL section of
// /root/reference/ffx-fsr/ffx_a.h + ffx_fsr1.h can be compiled *verbatim* as host C++17
// (see oracle/build_ref.sh).  The shim supplies language semantics only (vector types with
// the swizzles the headers use, component-wise operators, GLSL built-ins, an emulated
// float16_t); every arithmetic token of FsrEasuF/H and FsrRcasF/H is the reference's own.
//
// Pinned semantics where GLSL/HLSL leave latitude:
//   * no FMA contraction (build flag -ffp-contract=off)
//   * min/max = IEEE-754 minNum/maxNum (a NaN operand loses) with -0 < +0: what v_min_f32 / v_max_f32 and their
//     _f16 forms do on the GPUs the reference's shaders run on (AMD GCN3 ISA "V_MIN_F32"; PTX min.f32 likewise), where
//     C's fmin / fmax may return either zero
//   * 1.0/x is the correctly rounded IEEE division
//   * float16_t: every operation is computed exactly-enough in double and rounded once to
//     binary16 round-to-nearest-even (denormals kept, overflow -> inf)
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace glsl {

typedef unsigned int uint;

// ------------------------------------------------------------------------------------------
// binary16 helpers
// ------------------------------------------------------------------------------------------
static inline double round_to_half(double v) {
  if (std::isnan(v) || std::isinf(v) || v == 0.0) return v;
  double a = std::fabs(v);
  if (a >= 65520.0) return v < 0 ? -INFINITY : INFINITY;  // rounds up past max finite
  int e;
  std::frexp(a, &e);            // a = m * 2^e, m in [0.5,1)
  int q = e - 11;               // quantum exponent for 11 significant bits
  if (q < -24) q = -24;         // subnormal quantum 2^-24
  double s = std::ldexp(1.0, -q);
  double r = std::nearbyint(a * s);  // default rounding mode: nearest even; a*s exact
  r = std::ldexp(r, q);
  return v < 0 ? -r : r;
}
static inline uint16_t half_bits_from_value(double v) {  // v already representable
  uint16_t sign = std::signbit(v) ? 0x8000u : 0u;
  if (std::isnan(v)) return (uint16_t)(sign | 0x7e00u);
  double a = std::fabs(v);
  if (std::isinf(a)) return (uint16_t)(sign | 0x7c00u);
  if (a == 0.0) return sign;
  int e;
  double m = std::frexp(a, &e);  // a = m*2^e
  int E = e - 1;                 // a = (2m)*2^E, 2m in [1,2)
  if (E < -14) {                 // subnormal
    return (uint16_t)(sign | (uint16_t)std::ldexp(a, 24));
  }
  uint16_t mant = (uint16_t)std::ldexp(2.0 * m - 1.0, 10);
  return (uint16_t)(sign | ((E + 15) << 10) | mant);
}
static inline double half_value_from_bits(uint16_t h) {
  int s = (h >> 15) & 1, e = (h >> 10) & 31, m = h & 1023;
  double v;
  if (e == 0) v = std::ldexp((double)m, -24);
  else if (e == 31) v = m ? NAN : INFINITY;
  else v = std::ldexp(1.0 + m / 1024.0, e - 15);
  return s ? -v : v;
}

struct float16_t {
  float v;  // always holds a binary16-representable value
  float16_t() : v(0.f) {}
  float16_t(double d) : v((float)round_to_half(d)) {}
  float16_t(float d) : v((float)round_to_half((double)d)) {}
  float16_t(int d) : v((float)round_to_half((double)d)) {}
  float16_t(uint d) : v((float)round_to_half((double)d)) {}
  explicit operator float() const { return v; }
  explicit operator double() const { return v; }
  explicit operator uint() const { return (uint)v; }
  explicit operator int() const { return (int)v; }
};
#define H16_BIN(op) \
  static inline float16_t operator op(float16_t a, float16_t b) { return float16_t((double)a.v op (double)b.v); } \
  static inline float16_t& operator op##=(float16_t& a, float16_t b) { a = a op b; return a; }
H16_BIN(+) H16_BIN(-) H16_BIN(*) H16_BIN(/)
#undef H16_BIN
static inline float16_t operator-(float16_t a) { float16_t r; r.v = -a.v; return r; }
#define H16_CMP(op) static inline bool operator op(float16_t a, float16_t b) { return a.v op b.v; }
H16_CMP(<) H16_CMP(>) H16_CMP(<=) H16_CMP(>=) H16_CMP(==) H16_CMP(!=)
#undef H16_CMP

// ------------------------------------------------------------------------------------------
// vectors with the swizzles the reference headers use
// ------------------------------------------------------------------------------------------
template <class T> struct v2;
template <class T> struct v3;
template <class T> struct v4;

// Swizzle proxies: plain aggregates overlaying the component array of the owning vector.
template <class T, int A, int B> struct sw2 {
  T d[4];
  operator v2<T>() const;
  sw2& operator=(const v2<T>& o);
};
template <class T, int A, int B, int C> struct sw3 {
  T d[4];
  operator v3<T>() const;
  sw3& operator=(const v3<T>& o);
};

template <class T> struct v2 {
  union {
    T d[2];
    struct { T x, y; };
    struct { T r, g; };
    sw2<T, 0, 0> xx; sw2<T, 1, 1> yy; sw2<T, 0, 1> xy; sw2<T, 0, 1> rg;
    sw3<T, 0, 0, 0> xxx; sw3<T, 1, 1, 1> yyy;
  };
  v2() : d{T(), T()} {}
  v2(T a, T b) : d{a, b} {}
  template <class U> explicit v2(const v2<U>& o) : d{T(o.x), T(o.y)} {}
  template <class U, int A, int B> explicit v2(const sw2<U, A, B>& o) : d{T(o.d[A]), T(o.d[B])} {}
  T& operator[](int i) { return d[i]; }
  const T& operator[](int i) const { return d[i]; }
};
template <class T> struct v3 {
  union {
    T d[3];
    struct { T x, y, z; };
    struct { T r, g, b; };
    sw2<T, 0, 0> xx; sw2<T, 1, 1> yy; sw2<T, 2, 2> zz; sw2<T, 0, 1> xy; sw2<T, 1, 2> yz;
    sw3<T, 0, 0, 0> xxx; sw3<T, 1, 1, 1> yyy; sw3<T, 2, 2, 2> zzz; sw3<T, 0, 1, 2> xyz; sw3<T, 0, 1, 2> rgb;
  };
  v3() : d{T(), T(), T()} {}
  v3(T a, T b, T c) : d{a, b, c} {}
  template <class U> explicit v3(const v3<U>& o) : d{T(o.x), T(o.y), T(o.z)} {}
  template <class U, int A, int B, int C> explicit v3(const sw3<U, A, B, C>& o) : d{T(o.d[A]), T(o.d[B]), T(o.d[C])} {}
  T& operator[](int i) { return d[i]; }
  const T& operator[](int i) const { return d[i]; }
};
template <class T> struct v4 {
  union {
    T d[4];
    struct { T x, y, z, w; };
    struct { T r, g, b, a; };
    sw2<T, 0, 0> xx; sw2<T, 1, 1> yy; sw2<T, 2, 2> zz; sw2<T, 3, 3> ww;
    sw2<T, 0, 1> xy; sw2<T, 2, 3> zw; sw2<T, 1, 2> yz;
    sw3<T, 0, 0, 0> xxx; sw3<T, 1, 1, 1> yyy; sw3<T, 2, 2, 2> zzz; sw3<T, 0, 1, 2> xyz; sw3<T, 0, 1, 2> rgb;
  };
  v4() : d{T(), T(), T(), T()} {}
  v4(T a, T b, T c, T e) : d{a, b, c, e} {}
  v4(const v3<T>& o, T e) : d{o.x, o.y, o.z, e} {}
  v4(const v2<T>& p, const v2<T>& q) : d{p.x, p.y, q.x, q.y} {}
  template <class U> explicit v4(const v4<U>& o) : d{T(o.x), T(o.y), T(o.z), T(o.w)} {}
  T& operator[](int i) { return d[i]; }
  const T& operator[](int i) const { return d[i]; }
};
template <class T, int A, int B> sw2<T, A, B>::operator v2<T>() const { return v2<T>(d[A], d[B]); }
template <class T, int A, int B> sw2<T, A, B>& sw2<T, A, B>::operator=(const v2<T>& o) { d[A] = o.x; d[B] = o.y; return *this; }
template <class T, int A, int B, int C> sw3<T, A, B, C>::operator v3<T>() const { return v3<T>(d[A], d[B], d[C]); }
template <class T, int A, int B, int C> sw3<T, A, B, C>& sw3<T, A, B, C>::operator=(const v3<T>& o) { d[A] = o.x; d[B] = o.y; d[C] = o.z; return *this; }

typedef v2<float> vec2;   typedef v3<float> vec3;   typedef v4<float> vec4;
typedef v2<uint> uvec2;   typedef v3<uint> uvec3;   typedef v4<uint> uvec4;
typedef v2<int> ivec2;    typedef v3<int> ivec3;    typedef v4<int> ivec4;
typedef v2<bool> bvec2;   typedef v3<bool> bvec3;   typedef v4<bool> bvec4;
typedef v2<float16_t> f16vec2; typedef v3<float16_t> f16vec3; typedef v4<float16_t> f16vec4;
typedef v2<uint16_t> u16vec2;  typedef v3<uint16_t> u16vec3;  typedef v4<uint16_t> u16vec4;
typedef v2<int16_t> i16vec2;   typedef v3<int16_t> i16vec3;   typedef v4<int16_t> i16vec4;

// Non-template component-wise operators, one set per concrete type, so that swizzle proxies
// convert implicitly at call sites.
#define VEC_MAP2(V, e) V((e(0)), (e(1)))
#define VEC_MAP3(V, e) V((e(0)), (e(1)), (e(2)))
#define VEC_MAP4(V, e) V((e(0)), (e(1)), (e(2)), (e(3)))
#define DEF_BINOP(V, S, N, op)                                                                              \
  static inline V operator op(const V& a, const V& b) { V r; for (int i = 0; i < N; ++i) r.d[i] = (S)(a.d[i] op b.d[i]); return r; } \
  static inline V operator op(const V& a, S b) { V r; for (int i = 0; i < N; ++i) r.d[i] = (S)(a.d[i] op b); return r; }            \
  static inline V operator op(S a, const V& b) { V r; for (int i = 0; i < N; ++i) r.d[i] = (S)(a op b.d[i]); return r; }            \
  static inline V& operator op##=(V& a, const V& b) { for (int i = 0; i < N; ++i) a.d[i] = (S)(a.d[i] op b.d[i]); return a; }       \
  static inline V& operator op##=(V& a, S b) { for (int i = 0; i < N; ++i) a.d[i] = (S)(a.d[i] op b); return a; }
#define DEF_ARITH(V, S, N) DEF_BINOP(V, S, N, +) DEF_BINOP(V, S, N, -) DEF_BINOP(V, S, N, *) DEF_BINOP(V, S, N, /) \
  static inline V operator-(const V& a) { V r; for (int i = 0; i < N; ++i) r.d[i] = (S)(-a.d[i]); return r; }
#define DEF_BITS(V, S, N) DEF_BINOP(V, S, N, &) DEF_BINOP(V, S, N, |) DEF_BINOP(V, S, N, ^) DEF_BINOP(V, S, N, >>) DEF_BINOP(V, S, N, <<) \
  static inline V operator~(const V& a) { V r; for (int i = 0; i < N; ++i) r.d[i] = (S)(~a.d[i]); return r; } \
  static inline bool operator!=(const V& a, const V& b) { bool r = false; for (int i = 0; i < N; ++i) r = r || (a.d[i] != b.d[i]); return r; }
#define DEF_FAMILY_F(P, S) DEF_ARITH(P##2, S, 2) DEF_ARITH(P##3, S, 3) DEF_ARITH(P##4, S, 4)
#define DEF_FAMILY_I(P, S) DEF_FAMILY_F(P, S) DEF_BITS(P##2, S, 2) DEF_BITS(P##3, S, 3) DEF_BITS(P##4, S, 4)
DEF_FAMILY_F(vec, float)
DEF_FAMILY_F(f16vec, float16_t)
DEF_FAMILY_I(uvec, uint)
DEF_FAMILY_I(ivec, int)
DEF_FAMILY_I(u16vec, uint16_t)
DEF_FAMILY_I(i16vec, int16_t)

// ------------------------------------------------------------------------------------------
// built-ins (declared in this namespace so that unqualified calls from the headers bind here)
// ------------------------------------------------------------------------------------------
static inline float abs(float a) { return std::fabs(a); }
static inline int abs(int a) { return a < 0 ? -a : a; }
static inline int16_t abs(int16_t a) { return (int16_t)(a < 0 ? -a : a); }
static inline float16_t abs(float16_t a) { float16_t r; r.v = std::fabs(a.v); return r; }
template <class T> static inline T min_num(T a, T b) { return (a == 0 && b == 0) ? (std::signbit(a) ? a : b) : std::fmin(a, b); }  // -0 < +0
template <class T> static inline T max_num(T a, T b) { return (a == 0 && b == 0) ? (std::signbit(a) ? b : a) : std::fmax(a, b); }
static inline float min(float a, float b) { return min_num(a, b); }
static inline float max(float a, float b) { return max_num(a, b); }
static inline float16_t min(float16_t a, float16_t b) { float16_t r; r.v = min_num(a.v, b.v); return r; }
static inline float16_t max(float16_t a, float16_t b) { float16_t r; r.v = max_num(a.v, b.v); return r; }
static inline uint min(uint a, uint b) { return a < b ? a : b; }
static inline uint max(uint a, uint b) { return a > b ? a : b; }
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline float floor(float a) { return std::floor(a); }
static inline float16_t floor(float16_t a) { return float16_t(std::floor((double)a.v)); }
static inline float sqrt(float a) { return std::sqrt(a); }
static inline float16_t sqrt(float16_t a) { return float16_t(std::sqrt((double)a.v)); }
static inline float pow(float a, float b) { return std::pow(a, b); }
static inline float16_t pow(float16_t a, float16_t b) { return float16_t(std::pow((double)a.v, (double)b.v)); }
static inline float trunc(float a) { return std::trunc(a); }
static inline float exp2(float a) { return std::exp2(a); }
static inline float log2(float a) { return std::log2(a); }
static inline float sin(float a) { return std::sin(a); }
static inline float cos(float a) { return std::cos(a); }
static inline float fract(float a) { return a - std::floor(a); }
static inline float16_t fract(float16_t a) { return a - floor(a); }
static inline float clamp(float x, float lo, float hi) { return min(max(x, lo), hi); }
static inline float16_t clamp(float16_t x, float16_t lo, float16_t hi) { return min(max(x, lo), hi); }
static inline float mix(float x, float y, float a) { return x * (1.0f - a) + y * a; }
static inline float16_t mix(float16_t x, float16_t y, float16_t a) { return x * (float16_t(1.0) - a) + y * a; }

#define DEF_FN1(V, N, fn) static inline V fn(const V& a) { V r; for (int i = 0; i < N; ++i) r.d[i] = fn(a.d[i]); return r; }
#define DEF_FN2(V, N, fn) static inline V fn(const V& a, const V& b) { V r; for (int i = 0; i < N; ++i) r.d[i] = fn(a.d[i], b.d[i]); return r; }
#define DEF_FN3(V, N, fn) static inline V fn(const V& a, const V& b, const V& c) { V r; for (int i = 0; i < N; ++i) r.d[i] = fn(a.d[i], b.d[i], c.d[i]); return r; }
#define DEF_FLOATFNS(V, N) DEF_FN1(V, N, abs) DEF_FN1(V, N, floor) DEF_FN1(V, N, sqrt) DEF_FN1(V, N, fract) \
  DEF_FN2(V, N, min) DEF_FN2(V, N, max) DEF_FN2(V, N, pow) DEF_FN3(V, N, clamp) DEF_FN3(V, N, mix)
DEF_FLOATFNS(vec2, 2) DEF_FLOATFNS(vec3, 3) DEF_FLOATFNS(vec4, 4)
DEF_FLOATFNS(f16vec2, 2) DEF_FLOATFNS(f16vec3, 3) DEF_FLOATFNS(f16vec4, 4)
DEF_FN1(vec2, 2, sin) DEF_FN1(vec3, 3, sin) DEF_FN1(vec4, 4, sin)
DEF_FN1(vec2, 2, cos) DEF_FN1(vec3, 3, cos) DEF_FN1(vec4, 4, cos)
#define DEF_INTFNS(V, N) DEF_FN2(V, N, min) DEF_FN2(V, N, max)
DEF_INTFNS(uvec2, 2) DEF_INTFNS(uvec3, 3) DEF_INTFNS(uvec4, 4)
DEF_INTFNS(ivec2, 2) DEF_INTFNS(ivec3, 3) DEF_INTFNS(ivec4, 4)
static inline uint16_t min(uint16_t a, uint16_t b) { return a < b ? a : b; }
static inline uint16_t max(uint16_t a, uint16_t b) { return a > b ? a : b; }
static inline int16_t min(int16_t a, int16_t b) { return a < b ? a : b; }
static inline int16_t max(int16_t a, int16_t b) { return a > b ? a : b; }
DEF_INTFNS(u16vec2, 2) DEF_INTFNS(u16vec3, 3) DEF_INTFNS(u16vec4, 4)
DEF_INTFNS(i16vec2, 2) DEF_INTFNS(i16vec3, 3) DEF_INTFNS(i16vec4, 4)
DEF_FN1(ivec2, 2, abs) DEF_FN1(ivec3, 3, abs) DEF_FN1(ivec4, 4, abs)
DEF_FN1(i16vec2, 2, abs) DEF_FN1(i16vec3, 3, abs) DEF_FN1(i16vec4, 4, abs)

static inline float dot(const vec2& a, const vec2& b) { return a.x * b.x + a.y * b.y; }
static inline float dot(const vec3& a, const vec3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline float dot(const vec4& a, const vec4& b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
static inline float16_t dot(const f16vec2& a, const f16vec2& b) { return a.x * b.x + a.y * b.y; }
static inline float16_t dot(const f16vec3& a, const f16vec3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// bit casts
static inline float uintBitsToFloat(uint u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline uint floatBitsToUint(float f) { uint u; std::memcpy(&u, &f, 4); return u; }
#define DEF_CAST(VF, VU, N) \
  static inline VF uintBitsToFloat(const VU& u) { VF r; for (int i = 0; i < N; ++i) r.d[i] = uintBitsToFloat(u.d[i]); return r; } \
  static inline VU floatBitsToUint(const VF& f) { VU r; for (int i = 0; i < N; ++i) r.d[i] = floatBitsToUint(f.d[i]); return r; }
DEF_CAST(vec2, uvec2, 2) DEF_CAST(vec3, uvec3, 3) DEF_CAST(vec4, uvec4, 4)
static inline uint16_t halfBitsToUint16(float16_t h) { return half_bits_from_value((double)h.v); }
static inline float16_t uint16BitsToHalf(uint16_t u) { float16_t r; r.v = (float)half_value_from_bits(u); return r; }
#define DEF_HCAST(VH, VW, N) \
  static inline VW halfBitsToUint16(const VH& h) { VW r; for (int i = 0; i < N; ++i) r.d[i] = halfBitsToUint16(h.d[i]); return r; } \
  static inline VH uint16BitsToHalf(const VW& u) { VH r; for (int i = 0; i < N; ++i) r.d[i] = uint16BitsToHalf(u.d[i]); return r; }
DEF_HCAST(f16vec2, u16vec2, 2) DEF_HCAST(f16vec3, u16vec3, 3) DEF_HCAST(f16vec4, u16vec4, 4)

static inline uint bitfieldExtract(uint v, int off, int bits) { return bits >= 32 ? (v >> off) : ((v >> off) & ((1u << bits) - 1u)); }
static inline uint bitfieldInsert(uint base, uint ins, int off, int bits) {
  uint m = (bits >= 32 ? 0xffffffffu : ((1u << bits) - 1u)) << off;
  return (base & ~m) | ((ins << off) & m);
}
// pack/unpack (present only so the rest of the header compiles; half packs use RTNE)
static inline uint packHalf2x16(const vec2& a) { return (uint)half_bits_from_value(round_to_half(a.x)) | ((uint)half_bits_from_value(round_to_half(a.y)) << 16); }
static inline vec2 unpackHalf2x16(uint u) { return vec2((float)half_value_from_bits((uint16_t)(u & 0xffff)), (float)half_value_from_bits((uint16_t)(u >> 16))); }
static inline uint packUnorm2x16(const vec2& a) { return (uint)std::nearbyint(clamp(a.x, 0.f, 1.f) * 65535.f) | ((uint)std::nearbyint(clamp(a.y, 0.f, 1.f) * 65535.f) << 16); }
static inline vec2 unpackUnorm2x16(uint u) { return vec2((u & 0xffff) / 65535.f, (u >> 16) / 65535.f); }
static inline uint packUnorm4x8(const vec4& a) { uint r = 0; for (int i = 0; i < 4; ++i) r |= (uint)std::nearbyint(clamp(a.d[i], 0.f, 1.f) * 255.f) << (8 * i); return r; }
static inline vec4 unpackUnorm4x8(uint u) { return vec4((u & 255) / 255.f, ((u >> 8) & 255) / 255.f, ((u >> 16) & 255) / 255.f, (u >> 24) / 255.f); }
static inline f16vec2 unpackFloat2x16(uint u) { return f16vec2(uint16BitsToHalf((uint16_t)(u & 0xffff)), uint16BitsToHalf((uint16_t)(u >> 16))); }
static inline uint packFloat2x16(const f16vec2& h) { return (uint)halfBitsToUint16(h.x) | ((uint)halfBitsToUint16(h.y) << 16); }
static inline u16vec2 unpackUint2x16(uint u) { return u16vec2((uint16_t)(u & 0xffff), (uint16_t)(u >> 16)); }
static inline uint packUint2x16(const u16vec2& w) { return (uint)w.x | ((uint)w.y << 16); }
static inline uint64_t pack64(const uvec2& u) { return (uint64_t)u.x | ((uint64_t)u.y << 32); }
static inline uvec2 unpack32(uint64_t v) { return uvec2((uint)v, (uint)(v >> 32)); }
static inline u16vec4 unpackUint4x16(uint64_t v) { return u16vec4((uint16_t)v, (uint16_t)(v >> 16), (uint16_t)(v >> 32), (uint16_t)(v >> 48)); }
static inline uint64_t packUint4x16(const u16vec4& w) { return (uint64_t)w.x | ((uint64_t)w.y << 16) | ((uint64_t)w.z << 32) | ((uint64_t)w.w << 48); }

}  // namespace glsl
