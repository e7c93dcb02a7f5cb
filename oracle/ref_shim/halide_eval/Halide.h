// oracle/ref_shim/halide_eval/Halide.h — TEST INFRASTRUCTURE (oracle/_ref). A stand-in for "Halide.h" that EXECUTES a generator
// instead of compiling it: the reference's camera_isp/CameraIspGen.cpp is compiled from /root/reference as it lies over this
// header (oracle/ref_ispgen.cpp, `make -C oracle ref`), its main() builds the four pipelines out of Func / Var / Expr / RDom
// exactly as it would for Halide, compile_to_static_library() files them under the generated function's name, and a call of
// that function evaluates the output Func point by point (lazily, every Func memoised per coordinate).
//
// It implements the subset of the Halide front end that generator uses (release of late 2016, the buffer_t ABI):
//   Var, RDom / RVar, Expr arithmetic / comparison / logic with Halide's implicit-conversion rules for C++ literals, select
//   (3- and 5-argument), clamp, cast<>, absd, exp, pow, sum (inline reduction), undef<>, Func with one pure definition and
//   update definitions (constant or RDom coordinates in the updated dimensions, pure Vars elsewhere), ImageParam, Param<>,
//   BoundaryConditions::mirror_image / mirror_interior, Argument, Target; every SCHEDULING call is accepted and ignored —
//   a schedule never changes a Halide pipeline's values.
//
// Arithmetic, stated once (what a real Halide build may do differently at rounding level is listed in oracle/isp_pipe.h):
//   * Float(32) arithmetic is IEEE float, one rounding per operation, in the written association; nothing is contracted
//     into an FMA (build with -ffp-contract=off);
//   * a float division by a CONSTANT is a multiplication by the constant's float reciprocal (the simplifier of that Halide:
//     "x / 2 -> x * 0.5"); a division by anything else is a division. (HALIDE_EVAL_TRUE_DIVISION=1 in the environment keeps the
//     division, to measure how much of an output depends on this reading: tests/test_cpu_isp.py);
//   * integer % and / are Halide's: the remainder takes the sign of the divisor (never negative for a positive one),
//     division rounds toward minus infinity;
//   * float -> integer casts truncate toward zero (C semantics; every such cast in the generator follows a clamp);
//   * exp / pow on Float(32) are the C library's expf / powf (Halide's exp_f32 / pow_f32 call them);
//   * min / max / clamp on floats: clamp(a, lo, hi) = max(min(a, hi), lo) with `<` selections.
// Nothing here is performance code: a 128 x 96 image through the full pipeline takes a second or two.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "halide_buffer_t.h"

namespace Halide {

// ---- types and values ------------------------------------------------------------------------------------------------
struct Type {
  enum Code { IntC, UIntC, FloatC } code;
  int bits;
  bool is_float() const { return code == FloatC; }
  bool is_int() const { return code == IntC; }
  bool is_uint() const { return code == UIntC; }
  bool is_bool() const { return code == UIntC && bits == 1; }
  bool operator==(const Type& o) const { return code == o.code && bits == o.bits; }
  bool operator!=(const Type& o) const { return !(*this == o); }
};
inline Type Int(int bits) { return Type{Type::IntC, bits}; }
inline Type UInt(int bits) { return Type{Type::UIntC, bits}; }
inline Type Float(int bits) { return Type{Type::FloatC, bits}; }
inline Type Bool() { return UInt(1); }
template <typename T> struct type_of_t;
template <> struct type_of_t<float> { static Type get() { return Float(32); } };
template <> struct type_of_t<int> { static Type get() { return Int(32); } };
template <> struct type_of_t<bool> { static Type get() { return Bool(); } };
template <> struct type_of_t<uint8_t> { static Type get() { return UInt(8); } };
template <> struct type_of_t<uint16_t> { static Type get() { return UInt(16); } };
template <> struct type_of_t<int16_t> { static Type get() { return Int(16); } };
template <> struct type_of_t<uint32_t> { static Type get() { return UInt(32); } };

namespace Internal {
[[noreturn]] inline void fail(const std::string& m) { throw std::runtime_error("halide_eval: " + m); }

struct Value {
  union { int64_t i; float f; };
  Value() : i(0) {}
};
inline int64_t wrap(int64_t v, Type t) {  // an integer value as type t holds it
  if (t.bits >= 64) return v;
  const uint64_t mask = (uint64_t(1) << t.bits) - 1;
  uint64_t u = uint64_t(v) & mask;
  if (t.is_int() && (u >> (t.bits - 1))) return int64_t(u | ~mask);
  return int64_t(u);
}
inline Value convert(Value v, Type from, Type to) {
  Value r;
  if (from.is_float()) {
    if (to.is_float()) r.f = v.f;
    else r.i = wrap((int64_t)v.f, to);  // truncation toward zero
  } else {
    if (to.is_float()) r.f = (float)v.i;
    else if (to.is_bool()) r.i = v.i != 0;
    else r.i = wrap(v.i, to);
  }
  return r;
}

struct Env {  // the bindings of one Func definition being evaluated: its pure Vars and the RVars of a reduction
  int n = 0;
  int ids[10];
  int64_t vals[10];
  void bind(int id, int64_t v) {
    for (int k = 0; k < n; ++k) if (ids[k] == id) { vals[k] = v; return; }
    if (n == 10) fail("too many variables in one definition");
    ids[n] = id; vals[n++] = v;
  }
  int64_t get(int id, const std::string& name) const {
    for (int k = 0; k < n; ++k) if (ids[k] == id) return vals[k];
    fail("variable " + name + " is not bound by the definition being evaluated");
  }
};

struct Node {
  Type type;
  explicit Node(Type t) : type(t) {}
  virtual ~Node() {}
  virtual Value eval(const Env& e) const = 0;
  virtual void visit(const std::function<void(const Node*)>& f) const { f(this); }
};
inline int next_id() { static int id = 0; return ++id; }
}  // namespace Internal

struct Expr {
  std::shared_ptr<const Internal::Node> n;
  Expr() {}
  Expr(std::shared_ptr<const Internal::Node> p) : n(std::move(p)) {}
  Expr(int v);
  Expr(float v);
  bool defined() const { return (bool)n; }
  Type type() const { if (!n) Internal::fail("undefined Expr"); return n->type; }
};

namespace Internal {
struct Imm : Node {
  Value v; bool literal_int;
  Imm(Type t, Value val, bool lit) : Node(t), v(val), literal_int(lit) {}
  Value eval(const Env&) const override { return v; }
};
struct VarNode : Node {
  int id; std::string name; bool reduction;
  VarNode(int i, std::string nm, bool r) : Node(Int(32)), id(i), name(std::move(nm)), reduction(r) {}
  Value eval(const Env& e) const override { Value r; r.i = e.get(id, name); return r; }
};
struct ParamSlot { Type type; Value v; std::string name; };
struct ParamNode : Node {
  std::shared_ptr<ParamSlot> s;
  explicit ParamNode(std::shared_ptr<ParamSlot> p) : Node(p->type), s(std::move(p)) {}
  Value eval(const Env&) const override { return s->v; }
};
struct UndefNode : Node {
  explicit UndefNode(Type t) : Node(t) {}
  Value eval(const Env&) const override { fail("undef evaluated"); }
};
struct CastNode : Node {
  Expr a;
  CastNode(Type t, Expr x) : Node(t), a(std::move(x)) {}
  Value eval(const Env& e) const override { return convert(a.n->eval(e), a.n->type, type); }
  void visit(const std::function<void(const Node*)>& f) const override { f(this); a.n->visit(f); }
};
enum Op { Add, Sub, Mul, Div, Mod, Min, Max, LT, LE, GT, GE, EQ, NE, And, Or, Absd };
struct BinNode : Node {
  Op op; Expr a, b;  // operand types equal (match_types ran at construction)
  BinNode(Type t, Op o, Expr x, Expr y) : Node(t), op(o), a(std::move(x)), b(std::move(y)) {}
  Value eval(const Env& e) const override {
    const Value x = a.n->eval(e), y = b.n->eval(e);
    const Type ot = a.n->type;
    Value r;
    if (ot.is_float()) {
      switch (op) {
        case Add: r.f = x.f + y.f; break;
        case Sub: r.f = x.f - y.f; break;
        case Mul: r.f = x.f * y.f; break;
        case Div: r.f = x.f / y.f; break;
        case Min: r.f = y.f < x.f ? y.f : x.f; break;
        case Max: r.f = x.f < y.f ? y.f : x.f; break;
        case Absd: r.f = x.f < y.f ? y.f - x.f : x.f - y.f; break;
        case LT: r.i = x.f < y.f; break;
        case LE: r.i = x.f <= y.f; break;
        case GT: r.i = x.f > y.f; break;
        case GE: r.i = x.f >= y.f; break;
        case EQ: r.i = x.f == y.f; break;
        case NE: r.i = x.f != y.f; break;
        default: fail("operator not defined on floats");
      }
      return r;
    }
    switch (op) {
      case Add: r.i = wrap(x.i + y.i, type); break;
      case Sub: r.i = wrap(x.i - y.i, type); break;
      case Mul: r.i = wrap(x.i * y.i, type); break;
      case Div: {
        if (y.i == 0) fail("integer division by zero");
        int64_t q = x.i / y.i;
        if ((x.i % y.i != 0) && ((x.i < 0) != (y.i < 0))) --q;
        r.i = wrap(q, type);
        break;
      }
      case Mod: {
        if (y.i == 0) fail("integer modulo by zero");
        int64_t m = x.i % y.i;
        if (m != 0 && ((m < 0) != (y.i < 0))) m += y.i;
        r.i = wrap(m, type);
        break;
      }
      case Min: r.i = y.i < x.i ? y.i : x.i; break;
      case Max: r.i = x.i < y.i ? y.i : x.i; break;
      case Absd: r.i = x.i < y.i ? y.i - x.i : x.i - y.i; break;
      case LT: r.i = x.i < y.i; break;
      case LE: r.i = x.i <= y.i; break;
      case GT: r.i = x.i > y.i; break;
      case GE: r.i = x.i >= y.i; break;
      case EQ: r.i = x.i == y.i; break;
      case NE: r.i = x.i != y.i; break;
      case And: r.i = (x.i != 0) && (y.i != 0); break;
      case Or: r.i = (x.i != 0) || (y.i != 0); break;
    }
    return r;
  }
  void visit(const std::function<void(const Node*)>& f) const override { f(this); a.n->visit(f); b.n->visit(f); }
};
struct NotNode : Node {
  Expr a;
  explicit NotNode(Expr x) : Node(Bool()), a(std::move(x)) {}
  Value eval(const Env& e) const override { Value r; r.i = a.n->eval(e).i == 0; return r; }
  void visit(const std::function<void(const Node*)>& f) const override { f(this); a.n->visit(f); }
};
struct SelectNode : Node {  // only the chosen branch is evaluated (no side effects exist, so this is Halide's value)
  Expr c, t, f_;
  SelectNode(Expr cc, Expr tt, Expr ff) : Node(tt.type()), c(std::move(cc)), t(std::move(tt)), f_(std::move(ff)) {}
  Value eval(const Env& e) const override { return c.n->eval(e).i ? t.n->eval(e) : f_.n->eval(e); }
  void visit(const std::function<void(const Node*)>& f) const override { f(this); c.n->visit(f); t.n->visit(f); f_.n->visit(f); }
};
struct MathNode : Node {
  int fn; Expr a, b;  // 0 exp, 1 pow
  MathNode(int k, Expr x, Expr y) : Node(Float(32)), fn(k), a(std::move(x)), b(std::move(y)) {}
  Value eval(const Env& e) const override {
    Value r;
    if (fn == 0) r.f = ::expf(a.n->eval(e).f);
    else r.f = ::powf(a.n->eval(e).f, b.n->eval(e).f);
    return r;
  }
  void visit(const std::function<void(const Node*)>& f) const override { f(this); a.n->visit(f); if (b.n) b.n->visit(f); }
};
inline Expr make_const(Type t, double v) {
  Value x;
  if (t.is_float()) x.f = (float)v;
  else {
    x.i = (int64_t)v;
    if (wrap(x.i, t) != x.i && !t.is_bool()) fail("a literal does not fit the type of the expression it is combined with");
    if (t.is_bool()) x.i = x.i != 0;
  }
  return Expr(std::make_shared<Imm>(t, x, false));
}
inline bool is_literal_int(const Expr& e) { auto p = dynamic_cast<const Imm*>(e.n.get()); return p && p->literal_int; }
inline bool is_const(const Expr& e) { return dynamic_cast<const Imm*>(e.n.get()) != nullptr; }
inline Expr cast_to(Type t, Expr a) {
  if (a.type() == t) return a;
  if (auto p = dynamic_cast<const Imm*>(a.n.get())) {  // fold, so that coerced literals stay constants
    return Expr(std::make_shared<Imm>(t, convert(p->v, p->type, t), false));
  }
  return Expr(std::make_shared<CastNode>(t, std::move(a)));
}
// Halide's match_types (IROperator.cpp) for scalars, preceded by what its (Expr, int) / (Expr, float) operator overloads do: a
// C++ int literal takes the other operand's type
inline void match_types(Expr& a, Expr& b) {
  if (a.type() == b.type()) return;
  if (is_literal_int(b)) { b = cast_to(a.type(), b); return; }
  if (is_literal_int(a)) { a = cast_to(b.type(), a); return; }
  const Type ta = a.type(), tb = b.type();
  if (!ta.is_float() && tb.is_float()) a = cast_to(tb, a);
  else if (ta.is_float() && !tb.is_float()) b = cast_to(ta, b);
  else if (ta.is_float() && tb.is_float()) { if (ta.bits < tb.bits) a = cast_to(tb, a); else b = cast_to(ta, b); }
  else if (ta.is_uint() && tb.is_uint()) { if (ta.bits < tb.bits) a = cast_to(tb, a); else b = cast_to(ta, b); }
  else { const int bits = ta.bits > tb.bits ? ta.bits : tb.bits; a = cast_to(Int(bits), a); b = cast_to(Int(bits), b); }
}
inline Expr binary(Op op, Expr a, Expr b) {
  match_types(a, b);
  const bool cmp = op >= LT && op <= NE;
  if (op == And || op == Or) { if (!a.type().is_bool() || !b.type().is_bool()) fail("&& / || need boolean operands"); }
  if (op == Div && a.type().is_float() && is_const(b) && !std::getenv("HALIDE_EVAL_TRUE_DIVISION")) {
    // the simplifier's "convert const float division to multiplication": x / c -> x * (1 / c)
    const float c = static_cast<const Imm*>(b.n.get())->v.f;
    return binary(Mul, a, make_const(Float(32), 1.0f / c));
  }
  const Type t = (cmp || op == And || op == Or) ? Bool() : a.type();
  return Expr(std::make_shared<BinNode>(t, op, std::move(a), std::move(b)));
}
}  // namespace Internal

inline Expr::Expr(int v) { Internal::Value x; x.i = v; n = std::make_shared<Internal::Imm>(Int(32), x, true); }
inline Expr::Expr(float v) { Internal::Value x; x.f = v; n = std::make_shared<Internal::Imm>(Float(32), x, false); }

#define HALIDE_EVAL_BINOP(sym, OP)                                                                      \
  inline Expr operator sym(Expr a, Expr b) { return Internal::binary(Internal::OP, std::move(a), std::move(b)); } \
  inline Expr operator sym(Expr a, int b) { return Internal::binary(Internal::OP, std::move(a), Expr(b)); }       \
  inline Expr operator sym(int a, Expr b) { return Internal::binary(Internal::OP, Expr(a), std::move(b)); }       \
  inline Expr operator sym(Expr a, float b) { return Internal::binary(Internal::OP, std::move(a), Expr(b)); }     \
  inline Expr operator sym(float a, Expr b) { return Internal::binary(Internal::OP, Expr(a), std::move(b)); }
HALIDE_EVAL_BINOP(+, Add)
HALIDE_EVAL_BINOP(-, Sub)
HALIDE_EVAL_BINOP(*, Mul)
HALIDE_EVAL_BINOP(/, Div)
HALIDE_EVAL_BINOP(%, Mod)
HALIDE_EVAL_BINOP(<, LT)
HALIDE_EVAL_BINOP(<=, LE)
HALIDE_EVAL_BINOP(>, GT)
HALIDE_EVAL_BINOP(>=, GE)
HALIDE_EVAL_BINOP(==, EQ)
HALIDE_EVAL_BINOP(!=, NE)
#undef HALIDE_EVAL_BINOP
inline Expr operator&&(Expr a, Expr b) { return Internal::binary(Internal::And, std::move(a), std::move(b)); }
inline Expr operator||(Expr a, Expr b) { return Internal::binary(Internal::Or, std::move(a), std::move(b)); }
inline Expr operator!(Expr a) { return Expr(std::make_shared<Internal::NotNode>(std::move(a))); }
inline Expr operator-(Expr a) { return Internal::binary(Internal::Sub, Internal::make_const(a.type(), 0), a); }
inline Expr& operator+=(Expr& a, Expr b) { a = a + b; return a; }
inline Expr& operator-=(Expr& a, Expr b) { a = a - b; return a; }
inline Expr& operator*=(Expr& a, Expr b) { a = a * b; return a; }
inline Expr& operator/=(Expr& a, Expr b) { a = a / b; return a; }

template <typename T> Expr cast(Expr a) { return Internal::cast_to(type_of_t<T>::get(), std::move(a)); }
inline Expr cast(Type t, Expr a) { return Internal::cast_to(t, std::move(a)); }
template <typename T> Expr undef() { return Expr(std::make_shared<Internal::UndefNode>(type_of_t<T>::get())); }
inline Expr min(Expr a, Expr b) { return Internal::binary(Internal::Min, std::move(a), std::move(b)); }
inline Expr max(Expr a, Expr b) { return Internal::binary(Internal::Max, std::move(a), std::move(b)); }
inline Expr absd(Expr a, Expr b) { return Internal::binary(Internal::Absd, std::move(a), std::move(b)); }
inline Expr clamp(Expr a, Expr lo, Expr hi) {  // IROperator.h: Max(Min(a, hi), lo) with the bounds cast to a's type
  lo = Internal::cast_to(a.type(), lo);
  hi = Internal::cast_to(a.type(), hi);
  return max(min(std::move(a), std::move(hi)), std::move(lo));
}
inline Expr select(Expr c, Expr t, Expr f) {  // IROperator.h: int literals take the other branch's type; no other coercion
  if (Internal::is_literal_int(t) && !Internal::is_literal_int(f)) t = Internal::cast_to(f.type(), t);
  if (Internal::is_literal_int(f) && !Internal::is_literal_int(t)) f = Internal::cast_to(t.type(), f);
  if (Internal::is_literal_int(c)) c = Internal::cast_to(Bool(), c);
  if (!c.type().is_bool()) Internal::fail("select: the condition is not boolean");
  if (t.type() != f.type()) Internal::fail("select: the branches differ in type");
  return Expr(std::make_shared<Internal::SelectNode>(std::move(c), std::move(t), std::move(f)));
}
inline Expr select(Expr c0, Expr v0, Expr c1, Expr v1, Expr v2) { return select(std::move(c0), std::move(v0), select(std::move(c1), std::move(v1), std::move(v2))); }
inline Expr exp(Expr a) {
  if (a.type() != Float(32)) a = Internal::cast_to(Float(32), a);
  return Expr(std::make_shared<Internal::MathNode>(0, std::move(a), Expr()));
}
inline Expr pow(Expr a, Expr b) {
  a = Internal::cast_to(Float(32), a);
  b = Internal::cast_to(Float(32), b);
  return Expr(std::make_shared<Internal::MathNode>(1, std::move(a), std::move(b)));
}

// ---- variables and reduction domains ---------------------------------------------------------------------------------
struct Var {
  std::shared_ptr<const Internal::VarNode> v;
  Var() : v(std::make_shared<Internal::VarNode>(Internal::next_id(), "v", false)) {}
  explicit Var(const std::string& name) : v(std::make_shared<Internal::VarNode>(Internal::next_id(), name, false)) {}
  operator Expr() const { return Expr(v); }
};
namespace Internal {
struct RDomImpl {
  std::vector<Expr> mins, extents;
  std::vector<std::shared_ptr<const VarNode>> vars;
};
struct RVarNode : VarNode {  // knows its domain, so that a definition / a sum() can find what it iterates over
  std::weak_ptr<RDomImpl> dom;
  RVarNode(int i, std::string nm) : VarNode(i, std::move(nm), true) {}
};
}  // namespace Internal
struct RVar {
  std::shared_ptr<const Internal::VarNode> v;
  operator Expr() const { return Expr(v); }
};
struct RDom {
  std::shared_ptr<Internal::RDomImpl> d;
  RVar x, y;
  void init(std::vector<Expr> mn, std::vector<Expr> ex) {
    d = std::make_shared<Internal::RDomImpl>();
    d->mins = std::move(mn);
    d->extents = std::move(ex);
    for (size_t k = 0; k < d->mins.size(); ++k) {
      auto n = std::make_shared<Internal::RVarNode>(Internal::next_id(), k == 0 ? "r.x" : "r.y");
      n->dom = d;
      d->vars.push_back(n);
    }
    x.v = d->vars[0];
    if (d->vars.size() > 1) y.v = d->vars[1];
  }
  RDom(Expr min, Expr extent) { init({std::move(min)}, {std::move(extent)}); }
  RDom(Expr min0, Expr extent0, Expr min1, Expr extent1) { init({std::move(min0), std::move(min1)}, {std::move(extent0), std::move(extent1)}); }
  operator Expr() const { if (d->vars.size() != 1) Internal::fail("a multi-dimensional RDom used as an Expr"); return Expr(d->vars[0]); }
};
namespace Internal {
// the reduction domain an expression (list) iterates over: at most one
inline std::shared_ptr<RDomImpl> find_rdom(const std::vector<Expr>& es) {
  std::shared_ptr<RDomImpl> found;
  for (const Expr& e : es)
    if (e.defined())
      e.n->visit([&](const Node* n) {
        if (auto r = dynamic_cast<const RVarNode*>(n)) {
          auto d = r->dom.lock();
          if (found && d != found) fail("two reduction domains in one definition");
          found = d;
        }
      });
  return found;
}
inline void for_each_rpoint(const std::shared_ptr<RDomImpl>& d, Env env, const std::function<void(const Env&)>& body) {
  if (!d) { body(env); return; }
  const Env outer = env;
  std::vector<int64_t> mn, ex;
  for (size_t k = 0; k < d->mins.size(); ++k) {
    mn.push_back(d->mins[k].n->eval(outer).i);
    ex.push_back(d->extents[k].n->eval(outer).i);
  }
  if (d->vars.size() == 1) {
    for (int64_t a = 0; a < ex[0]; ++a) { env.bind(d->vars[0]->id, mn[0] + a); body(env); }
  } else {  // x innermost, like Halide's loop nest over an RDom
    for (int64_t b = 0; b < ex[1]; ++b)
      for (int64_t a = 0; a < ex[0]; ++a) { env.bind(d->vars[0]->id, mn[0] + a); env.bind(d->vars[1]->id, mn[1] + b); body(env); }
  }
}
struct SumNode : Node {  // inline reduction: sum(e) over e's RDom, the other variables taken from the enclosing definition
  Expr a; std::shared_ptr<RDomImpl> dom;
  explicit SumNode(Expr x) : Node(x.type()), a(std::move(x)) { dom = find_rdom({a}); if (!dom) fail("sum() without an RDom"); }
  Value eval(const Env& e) const override {
    Value acc;
    if (type.is_float()) acc.f = 0.0f;
    for_each_rpoint(dom, e, [&](const Env& en) {
      const Value v = a.n->eval(en);
      if (type.is_float()) acc.f = acc.f + v.f; else acc.i = wrap(acc.i + v.i, type);
    });
    return acc;
  }
  // (its RVars are bound inside: they are not free variables of the enclosing definition, so visit() does not descend)
};
}  // namespace Internal
inline Expr sum(Expr e) { return Expr(std::make_shared<Internal::SumNode>(std::move(e))); }

// ---- Funcs ---------------------------------------------------------------------------------------------------------------
enum class TailStrategy { RoundUp, GuardWithIf, ShiftInwards, Auto };
struct Target {
  int natural_vector_size(Type t) const { return 256 / t.bits; }
};
inline Target get_target_from_environment() { return Target(); }
inline Target get_host_target() { return Target(); }

namespace Internal {
struct Key {
  int64_t c[4]; int n;
  bool operator==(const Key& o) const { return n == o.n && c[0] == o.c[0] && c[1] == o.c[1] && c[2] == o.c[2] && c[3] == o.c[3]; }
};
struct KeyHash {
  size_t operator()(const Key& k) const {
    uint64_t h = 1469598103934665603ull;
    for (int i = 0; i < 4; ++i) { h ^= (uint64_t)k.c[i] + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2); }
    return (size_t)h;
  }
};
struct FuncImpl;
inline std::vector<FuncImpl*>& all_funcs() { static std::vector<FuncImpl*> v; return v; }
struct Definition { std::vector<Expr> args; Expr value; std::shared_ptr<RDomImpl> dom; };
struct FuncImpl {
  std::string name;
  std::vector<std::shared_ptr<const VarNode>> pure_args;
  Expr pure_value;
  std::vector<Definition> updates;
  std::function<Value(const int64_t*, int)> native;  // boundary-conditioned images
  Type native_type = Int(32);
  std::vector<bool> pure_dim;  // with updates: the dimensions every update leaves to the pure Var
  std::unordered_map<Key, Value, KeyHash> memo;
  std::unordered_map<Key, int, KeyHash> column_state;  // 1 = being computed, 2 = done
  explicit FuncImpl(std::string n) : name(std::move(n)) { all_funcs().push_back(this); }
  ~FuncImpl() { auto& v = all_funcs(); for (size_t i = 0; i < v.size(); ++i) if (v[i] == this) { v.erase(v.begin() + i); break; } }
  bool defined() const { return pure_value.defined() || (bool)native; }
  Type type() const { if (native) return native_type; if (!pure_value.defined()) fail("Func " + name + " used before it is defined"); return pure_value.type(); }
  void clear() { memo.clear(); column_state.clear(); }

  Key column_of(const Key& k) const { Key c = k; for (int i = 0; i < k.n; ++i) if (!pure_dim[i]) c.c[i] = 0; return c; }
  Value eval(const Key& k) {
    if (native) return native(k.c, k.n);
    if ((int)pure_args.size() != k.n) fail("Func " + name + " called with the wrong number of arguments");
    if (updates.empty()) {
      auto it = memo.find(k);
      if (it != memo.end()) return it->second;
      Env e;
      for (int i = 0; i < k.n; ++i) e.bind(pure_args[i]->id, k.c[i]);
      const Value v = pure_value.n->eval(e);
      memo.emplace(k, v);
      return v;
    }
    // a read from inside the Func's own update definitions finds its column "being computed" and takes what has been written
    const Key col = column_of(k);
    if (column_state.find(col) == column_state.end()) { column_state[col] = 1; compute_column(k); column_state[col] = 2; }
    auto it = memo.find(k);
    if (it == memo.end()) fail("Func " + name + ": a value no definition has written is read (undef)");
    return it->second;
  }
  // all update definitions, in order, for the pure coordinates of k: every updated dimension is written by the updates' own
  // coordinates (constants or the RDom), which is the whole extent a realization would compute
  void compute_column(const Key& k) {
    Env base;
    for (int i = 0; i < k.n; ++i) if (pure_dim[i]) base.bind(pure_args[i]->id, k.c[i]);
    if (!dynamic_cast<const UndefNode*>(pure_value.n.get())) fail("Func " + name + ": updates over a defined pure step are not supported by this evaluator");
    for (const Definition& u : updates) {
      for_each_rpoint(u.dom, base, [&](const Env& e) {
        Key w; w.n = k.n; w.c[0] = w.c[1] = w.c[2] = w.c[3] = 0;
        for (int i = 0; i < k.n; ++i) w.c[i] = pure_dim[i] ? k.c[i] : u.args[i].n->eval(e).i;
        const Value v = convert(u.value.n->eval(e), u.value.type(), pure_value.type());
        memo[w] = v;
      });
    }
  }
};
struct CallNode : Node {
  std::shared_ptr<FuncImpl> f; std::vector<Expr> args;
  CallNode(std::shared_ptr<FuncImpl> fn, std::vector<Expr> a) : Node(fn->type()), f(std::move(fn)), args(std::move(a)) {}
  Value eval(const Env& e) const override {
    Key k; k.n = (int)args.size(); k.c[0] = k.c[1] = k.c[2] = k.c[3] = 0;
    for (int i = 0; i < k.n; ++i) k.c[i] = args[i].n->eval(e).i;
    return f->eval(k);
  }
  void visit(const std::function<void(const Node*)>& fn) const override { fn(this); for (const Expr& a : args) a.n->visit(fn); }
};
}  // namespace Internal

struct Stage {
  template <typename... A> Stage& reorder(A&&...) { return *this; }
  template <typename... A> Stage& unroll(A&&...) { return *this; }
  template <typename... A> Stage& vectorize(A&&...) { return *this; }
  template <typename... A> Stage& parallel(A&&...) { return *this; }
  template <typename... A> Stage& split(A&&...) { return *this; }
  template <typename... A> Stage& tile(A&&...) { return *this; }
};
struct OutputImageParam {
  OutputImageParam& set_stride(int, Expr) { return *this; }
  OutputImageParam& set_bounds(int, Expr, Expr) { return *this; }
  OutputImageParam& set_min(int, Expr) { return *this; }
  OutputImageParam& set_extent(int, Expr) { return *this; }
};

class Func;
class FuncRef {
  std::shared_ptr<Internal::FuncImpl> f;
  std::vector<Expr> args;
 public:
  FuncRef(std::shared_ptr<Internal::FuncImpl> fn, std::vector<Expr> a) : f(std::move(fn)), args(std::move(a)) {}
  operator Expr() const {
    return Expr(std::make_shared<Internal::CallNode>(f, args));
  }
  // f(x, y) = e: the pure definition when f has none and every argument is a distinct Var; an update definition otherwise
  FuncRef& operator=(Expr e) {
    using namespace Internal;
    bool all_vars = true;
    for (const Expr& a : args) { auto v = dynamic_cast<const VarNode*>(a.n.get()); if (!v || v->reduction) all_vars = false; }
    if (!f->pure_value.defined()) {
      if (!all_vars) fail("Func " + f->name + ": the first definition must be pure");
      for (const Expr& a : args) f->pure_args.push_back(std::static_pointer_cast<const VarNode>(a.n));
      f->pure_value = std::move(e);
      f->pure_dim.assign(args.size(), true);
      return *this;
    }
    if (args.size() != f->pure_args.size()) fail("Func " + f->name + ": update with another dimensionality");
    Definition d;
    d.args = args;
    d.value = std::move(e);
    std::vector<Expr> all = d.args;
    all.push_back(d.value);
    d.dom = find_rdom(all);
    for (size_t i = 0; i < args.size(); ++i) {
      auto v = dynamic_cast<const VarNode*>(args[i].n.get());
      if (!(v && !v->reduction && v->id == f->pure_args[i]->id)) f->pure_dim[i] = false;
    }
    f->updates.push_back(std::move(d));
    return *this;
  }
  FuncRef& operator=(const FuncRef& o) { return *this = Expr(o); }
  FuncRef& operator=(int v) { return *this = Expr(v); }
  FuncRef& operator=(float v) { return *this = Expr(v); }
  friend class Func;
};

class Func {
  std::shared_ptr<Internal::FuncImpl> f;
  static std::string fresh() { return "f" + std::to_string(Internal::next_id()); }
 public:
  Func() : f(std::make_shared<Internal::FuncImpl>(fresh())) {}
  explicit Func(const std::string& name) : f(std::make_shared<Internal::FuncImpl>(name)) {}
  explicit Func(std::shared_ptr<Internal::FuncImpl> p) : f(std::move(p)) {}
  const std::shared_ptr<Internal::FuncImpl>& impl() const { return f; }
  bool defined() const { return f->defined(); }
  template <typename... A> FuncRef operator()(A&&... a) const {
    std::vector<Expr> args{Expr(std::forward<A>(a))...};
    return FuncRef(f, std::move(args));
  }
  // scheduling: accepted, ignored
  template <typename... A> Func& compute_at(A&&...) { return *this; }
  template <typename... A> Func& store_at(A&&...) { return *this; }
  template <typename... A> Func& compute_root(A&&...) { return *this; }
  template <typename... A> Func& store_root(A&&...) { return *this; }
  template <typename... A> Func& vectorize(A&&...) { return *this; }
  template <typename... A> Func& unroll(A&&...) { return *this; }
  template <typename... A> Func& parallel(A&&...) { return *this; }
  template <typename... A> Func& reorder(A&&...) { return *this; }
  template <typename... A> Func& split(A&&...) { return *this; }
  template <typename... A> Func& tile(A&&...) { return *this; }
  template <typename... A> Func& fold_storage(A&&...) { return *this; }
  template <typename... A> Func& bound(A&&...) { return *this; }
  Stage update(int = 0) { return Stage(); }
  OutputImageParam output_buffer() { return OutputImageParam(); }
  void print_loop_nest() {}
  template <typename ArgsT> void compile_to_static_library(const std::string& file, const ArgsT& args, const std::string& fn_name, const Target& = Target());
  template <typename ArgsT> void compile_to_assembly(const std::string&, const ArgsT&, const Target& = Target()) {}
  template <typename ArgsT> void compile_to_assembly(const std::string&, const ArgsT&, const std::string&, const Target& = Target()) {}
};

// ---- parameters -----------------------------------------------------------------------------------------------------------
namespace Internal {
struct ImageSlot { Type type; int dims; std::string name; buffer_t buf; bool bound = false; };
inline Value load(const ImageSlot& s, const int64_t* idx, int n) {
  if (!s.bound) fail("image parameter " + s.name + " read before it is bound");
  if (n != s.dims) fail("image parameter " + s.name + " indexed with the wrong number of coordinates");
  int64_t off = 0;
  for (int d = 0; d < n; ++d) {
    const int64_t r = idx[d] - s.buf.min[d];
    if (r < 0 || r >= s.buf.extent[d]) fail("image parameter " + s.name + " read out of bounds");
    off += r * s.buf.stride[d];
  }
  const uint8_t* p = s.buf.host + off * s.buf.elem_size;
  Value v;
  if (s.type.is_float()) { std::memcpy(&v.f, p, 4); return v; }
  switch (s.type.bits) {
    case 8: v.i = s.type.is_int() ? (int64_t) * reinterpret_cast<const int8_t*>(p) : (int64_t)*p; break;
    case 16: { uint16_t t; std::memcpy(&t, p, 2); v.i = s.type.is_int() ? (int64_t)(int16_t)t : (int64_t)t; break; }
    case 32: { uint32_t t; std::memcpy(&t, p, 4); v.i = s.type.is_int() ? (int64_t)(int32_t)t : (int64_t)t; break; }
    default: fail("image element type not supported");
  }
  return v;
}
struct ImageNode : Node {
  std::shared_ptr<ImageSlot> s; std::vector<Expr> args;
  ImageNode(std::shared_ptr<ImageSlot> p, std::vector<Expr> a) : Node(p->type), s(std::move(p)), args(std::move(a)) {}
  Value eval(const Env& e) const override {
    int64_t idx[4];
    for (size_t i = 0; i < args.size(); ++i) idx[i] = convert(args[i].n->eval(e), args[i].type(), Int(32)).i;
    return load(*s, idx, (int)args.size());
  }
  void visit(const std::function<void(const Node*)>& fn) const override { fn(this); for (const Expr& a : args) a.n->visit(fn); }
};
}  // namespace Internal

class ImageParam {
  std::shared_ptr<Internal::ImageSlot> s;
 public:
  ImageParam(Type t, int dims, const std::string& name = "image") : s(std::make_shared<Internal::ImageSlot>()) { s->type = t; s->dims = dims; s->name = name; }
  const std::shared_ptr<Internal::ImageSlot>& slot() const { return s; }
  template <typename... A> Expr operator()(A&&... a) const {
    std::vector<Expr> args{Expr(std::forward<A>(a))...};
    return Expr(std::make_shared<Internal::ImageNode>(s, std::move(args)));
  }
};
template <typename T> class Param {
  std::shared_ptr<Internal::ParamSlot> s;
 public:
  Param() : s(std::make_shared<Internal::ParamSlot>()) { s->type = type_of_t<T>::get(); s->name = "param"; }
  explicit Param(const std::string& name) : s(std::make_shared<Internal::ParamSlot>()) { s->type = type_of_t<T>::get(); s->name = name; }
  const std::shared_ptr<Internal::ParamSlot>& slot() const { return s; }
  operator Expr() const { return Expr(std::make_shared<Internal::ParamNode>(s)); }
};
struct Argument {
  std::shared_ptr<Internal::ImageSlot> image;
  std::shared_ptr<Internal::ParamSlot> scalar;
  Argument(const ImageParam& p) : image(p.slot()) {}
  template <typename T> Argument(const Param<T>& p) : scalar(p.slot()) {}
};

// ---- boundary conditions (BoundaryConditions.cpp of that Halide, for an ImageParam: every dimension, the buffer's own bounds)
namespace BoundaryConditions {
namespace detail {
inline int64_t emod(int64_t a, int64_t b) { int64_t m = a % b; if (m < 0) m += b; return m; }
inline Func conditioned(const ImageParam& im, bool interior) {
  auto impl = std::make_shared<Internal::FuncImpl>(interior ? "mirror_interior" : "mirror_image");
  auto slot = im.slot();
  impl->native_type = slot->type;
  impl->native = [slot, interior](const int64_t* c, int n) {
    int64_t idx[4];
    if (!slot->bound) Internal::fail("image parameter " + slot->name + " read before it is bound");
    for (int d = 0; d < n; ++d) {
      const int64_t mn = slot->buf.min[d], extent = slot->buf.extent[d];
      int64_t coord = c[d] - mn;
      if (interior) {  // abcd|cba: coord % (2 * (extent - 1)), reflected about extent - 1
        const int64_t limit = extent - 1;
        if (limit == 0) coord = 0;
        else { coord = emod(coord, 2 * limit); coord = limit - std::llabs(coord - limit); }
      } else {  // abcd|dcba
        coord = emod(coord, 2 * extent);
        if (coord >= extent) coord = 2 * extent - 1 - coord;
      }
      coord += mn;
      if (coord < mn) coord = mn;
      if (coord > mn + extent - 1) coord = mn + extent - 1;
      idx[d] = coord;
    }
    return Internal::load(*slot, idx, n);
  };
  return Func(impl);
}
}  // namespace detail
inline Func mirror_image(const ImageParam& im) { return detail::conditioned(im, false); }
inline Func mirror_interior(const ImageParam& im) { return detail::conditioned(im, true); }
}  // namespace BoundaryConditions

// ---- "compilation": a pipeline filed under the name of the function Halide would generate; calling it evaluates it ------------
namespace Internal {
struct Pipeline { std::shared_ptr<FuncImpl> out; std::vector<Argument> args; };
inline std::map<std::string, Pipeline>& pipelines() { static std::map<std::string, Pipeline> p; return p; }
inline std::mutex& eval_mutex() { static std::mutex m; return m; }

// one generated-function call: args in the generator's order (buffers as buffer_t*, scalars as doubles), then the output buffer
struct CallArg { const buffer_t* buf; double scalar; };
inline int run_pipeline(const std::string& name, const std::vector<CallArg>& in, buffer_t* out) {
  std::lock_guard<std::mutex> g(eval_mutex());
  auto it = pipelines().find(name);
  if (it == pipelines().end()) fail("no pipeline named " + name);
  Pipeline& p = it->second;
  if (in.size() != p.args.size()) fail("pipeline " + name + " called with the wrong number of arguments");
  for (size_t i = 0; i < in.size(); ++i) {
    if (p.args[i].image) {
      if (!in[i].buf) fail("pipeline " + name + ": a buffer argument is missing");
      p.args[i].image->buf = *in[i].buf;
      p.args[i].image->bound = true;
      if (p.args[i].image->buf.elem_size != p.args[i].image->type.bits / 8) fail("pipeline " + name + ": a buffer's element size does not match its ImageParam");
    } else {
      ParamSlot& s = *p.args[i].scalar;
      if (s.type.is_float()) s.v.f = (float)in[i].scalar;
      else if (s.type.is_bool()) s.v.i = in[i].scalar != 0;
      else s.v.i = (int64_t)in[i].scalar;
    }
  }
  for (FuncImpl* f : all_funcs()) f->clear();
  const Type t = p.out->type();
  if (out->elem_size != t.bits / 8) fail("pipeline " + name + ": the output buffer's element size does not match");
  const int dims = (int)p.out->pure_args.size();
  Key k; k.n = dims;
  int64_t ext[4] = {1, 1, 1, 1};
  for (int d = 0; d < dims; ++d) ext[d] = out->extent[d];
  for (int64_t c3 = 0; c3 < ext[3]; ++c3)
    for (int64_t c2 = 0; c2 < ext[2]; ++c2)
      for (int64_t c1 = 0; c1 < ext[1]; ++c1)
        for (int64_t c0 = 0; c0 < ext[0]; ++c0) {
          const int64_t c[4] = {c0 + out->min[0], c1 + out->min[1], c2 + out->min[2], c3 + out->min[3]};
          for (int d = 0; d < 4; ++d) k.c[d] = d < dims ? c[d] : 0;
          const Value v = p.out->eval(k);
          int64_t off = 0;
          for (int d = 0; d < dims; ++d) off += (c[d] - out->min[d]) * out->stride[d];
          uint8_t* dst = out->host + off * out->elem_size;
          if (t.is_float()) std::memcpy(dst, &v.f, 4);
          else if (t.bits == 8) *dst = (uint8_t)v.i;
          else if (t.bits == 16) { const uint16_t u = (uint16_t)v.i; std::memcpy(dst, &u, 2); }
          else { const uint32_t u = (uint32_t)v.i; std::memcpy(dst, &u, 4); }
        }
  for (FuncImpl* f : all_funcs()) f->clear();
  return 0;
}
}  // namespace Internal

template <typename ArgsT>
void Func::compile_to_static_library(const std::string&, const ArgsT& args, const std::string& fn_name, const Target&) {
  Internal::Pipeline p;
  p.out = f;
  for (const Argument& a : args) p.args.push_back(a);
  Internal::pipelines()[fn_name] = std::move(p);
}

}  // namespace Halide
