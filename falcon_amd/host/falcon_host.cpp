// falcon_host.cpp -- see falcon_host.h.  Mirrors, for the assembly path only:
//   src/fem/solidmech/LinearElasticityModel.cpp:49-289, PhaseFractureModel.cpp:82-494,
//   PhaseFractureExtModel.cpp:200-480,1327-1341, PhaseFractureExtItModel.cpp:371-510,1335-1395,
//   src/materials/Material.cpp:212-247 (newMaterial) and the materials' configure() methods.
#include "falcon_host.h"

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstring>
#include <set>
#include <sstream>

namespace falcon_host {

// =============================================================================================
//   Properties
// =============================================================================================
const std::any* Properties::lookup(const std::string& key) const {
  const size_t dot = key.find('.');
  if (dot == std::string::npos) {
    auto it = d_->m.find(key);
    return it == d_->m.end() ? nullptr : &it->second;
  }
  auto it = d_->m.find(key.substr(0, dot));
  if (it == d_->m.end()) return nullptr;
  const Properties* sub = std::any_cast<Properties>(&it->second);
  return sub ? sub->lookup(key.substr(dot + 1)) : nullptr;
}

void Properties::setAny(const std::string& key, std::any v) {
  const size_t dot = key.find('.');
  if (dot == std::string::npos) { d_->m[key] = std::move(v); return; }
  makeProps(key.substr(0, dot)).setAny(key.substr(dot + 1), std::move(v));
}

bool Properties::contains(const std::string& key) const { return lookup(key) != nullptr; }

std::vector<std::string> Properties::keys() const {
  std::vector<std::string> k;
  for (auto& kv : d_->m) k.push_back(kv.first);
  return k;
}

namespace {
template <class T> struct TypeName { static const char* name() { return "object"; } };
template <> struct TypeName<double> { static const char* name() { return "float"; } };
template <> struct TypeName<int> { static const char* name() { return "integer"; } };
template <> struct TypeName<bool> { static const char* name() { return "boolean"; } };
template <> struct TypeName<std::string> { static const char* name() { return "string"; } };
}  // namespace

template <class T>
bool Properties::find(T& out, const std::string& key) const {
  const std::any* a = lookup(key);
  if (!a) return false;
  if (const T* v = std::any_cast<T>(a)) { out = *v; return true; }
  if constexpr (std::is_same<T, double>::value) {            // integer literals convert to float like in Jive
    if (const int* v = std::any_cast<int>(a)) { out = *v; return true; }
  }
  if constexpr (std::is_same<T, int>::value) {
    if (const double* v = std::any_cast<double>(a)) { if (*v == std::floor(*v)) { out = (int)*v; return true; } }
  }
  throw Error(context, "property `" + key + "' has the wrong type (expected " + TypeName<T>::name() + ")");
}

template <class T>
void Properties::get(T& out, const std::string& key) const {
  if (!find(out, key)) throw Error(context, "undefined property: `" + key + "'");
}

template bool Properties::find<double>(double&, const std::string&) const;
template bool Properties::find<int>(int&, const std::string&) const;
template bool Properties::find<bool>(bool&, const std::string&) const;
template bool Properties::find<std::string>(std::string&, const std::string&) const;
template bool Properties::find<std::vector<std::string>>(std::vector<std::string>&, const std::string&) const;
template bool Properties::find<std::vector<double>>(std::vector<double>&, const std::string&) const;
template bool Properties::find<Vector>(Vector&, const std::string&) const;
template bool Properties::find<std::shared_ptr<MatrixBuilder>>(std::shared_ptr<MatrixBuilder>&, const std::string&) const;
template bool Properties::find<std::shared_ptr<Mesh>>(std::shared_ptr<Mesh>&, const std::string&) const;
template bool Properties::find<std::shared_ptr<XDofSpace>>(std::shared_ptr<XDofSpace>&, const std::string&) const;
template void Properties::get<double>(double&, const std::string&) const;
template void Properties::get<int>(int&, const std::string&) const;
template void Properties::get<bool>(bool&, const std::string&) const;
template void Properties::get<std::string>(std::string&, const std::string&) const;
template void Properties::get<Vector>(Vector&, const std::string&) const;
template void Properties::get<std::shared_ptr<MatrixBuilder>>(std::shared_ptr<MatrixBuilder>&, const std::string&) const;
template void Properties::get<std::shared_ptr<Mesh>>(std::shared_ptr<Mesh>&, const std::string&) const;
template void Properties::get<std::shared_ptr<XDofSpace>>(std::shared_ptr<XDofSpace>&, const std::string&) const;

Properties Properties::getProps(const std::string& key) const {
  const std::any* a = lookup(key);
  const Properties* p = a ? std::any_cast<Properties>(a) : nullptr;
  if (!p) throw Error(context, "undefined property set: `" + key + "'");
  Properties r = *p; r.context = context.empty() ? key : context + "." + key;
  return r;
}
Properties Properties::findProps(const std::string& key) const {
  const std::any* a = lookup(key);
  const Properties* p = a ? std::any_cast<Properties>(a) : nullptr;
  Properties r = p ? *p : Properties();
  r.context = context.empty() ? key : context + "." + key;
  return r;
}
Properties Properties::makeProps(const std::string& key) {
  const std::any* a = lookup(key);
  if (a) { if (const Properties* p = std::any_cast<Properties>(a)) return *p; }
  Properties p; p.context = context.empty() ? key : context + "." + key;
  setAny(key, std::any(p));
  return p;
}

struct Parser {
  const std::string& s; size_t i = 0;
  explicit Parser(const std::string& t) : s(t) {}
  [[noreturn]] void fail(const std::string& m) {
    int line = 1; for (size_t k = 0; k < i && k < s.size(); k++) if (s[k] == '\n') line++;
    throw Error("Properties::parse", m + " (line " + std::to_string(line) + ")");
  }
  void ws() {
    for (;;) {
      while (i < s.size() && std::isspace((unsigned char)s[i])) i++;
      if (i + 1 < s.size() && s[i] == '/' && s[i + 1] == '/') { while (i < s.size() && s[i] != '\n') i++; continue; }
      if (i + 1 < s.size() && s[i] == '/' && s[i + 1] == '*') {
        size_t e = s.find("*/", i + 2); if (e == std::string::npos) fail("unterminated comment"); i = e + 2; continue;
      }
      break;
    }
  }
  std::string ident() {
    size_t b = i;
    while (i < s.size() && (std::isalnum((unsigned char)s[i]) || s[i] == '_' || s[i] == '.')) i++;
    if (b == i) fail("identifier expected");
    return s.substr(b, i - b);
  }
  std::string str() {
    i++; size_t b = i;
    while (i < s.size() && s[i] != '"') i++;
    if (i >= s.size()) fail("unterminated string");
    return s.substr(b, i++ - b);
  }
  std::any number() {
    size_t b = i; bool isInt = true;
    if (s[i] == '+' || s[i] == '-') i++;
    while (i < s.size() && (std::isdigit((unsigned char)s[i]) || s[i] == '.' || s[i] == 'e' || s[i] == 'E' ||
                            ((s[i] == '+' || s[i] == '-') && (s[i - 1] == 'e' || s[i - 1] == 'E')))) {
      if (!std::isdigit((unsigned char)s[i])) isInt = false;
      i++;
    }
    const std::string t = s.substr(b, i - b);
    if (t.empty() || t == "+" || t == "-") fail("number expected");
    if (isInt) return std::any((int)std::stol(t));
    return std::any(std::stod(t));
  }
  std::any value() {
    ws();
    if (i >= s.size()) fail("value expected");
    const char c = s[i];
    if (c == '"') {
      std::string v = str(); ws();
      if (i < s.size() && s[i] == '{') {                         // typed block:  key = "Type" { ... }
        Properties p = block(); p.set("type", v); return std::any(p);
      }
      return std::any(v);
    }
    if (c == '{') return std::any(block());
    if (c == '[') {
      i++; std::vector<std::string> sv; std::vector<double> dv; bool strs = false, any = false;
      for (;;) {
        ws(); if (i < s.size() && s[i] == ']') { i++; break; }
        if (s[i] == '"') { sv.push_back(str()); strs = true; }
        else { std::any n = number(); dv.push_back(n.type() == typeid(int) ? std::any_cast<int>(n) : std::any_cast<double>(n)); }
        any = true; ws();
        if (i < s.size() && s[i] == ',') i++;
      }
      (void)any;
      return strs ? std::any(sv) : std::any(dv);
    }
    if (std::isalpha((unsigned char)c)) {
      std::string w = ident();
      if (w == "true") return std::any(true);
      if (w == "false") return std::any(false);
      fail("unexpected word `" + w + "'");
    }
    return number();
  }
  Properties block() {
    i++; Properties p = body('}'); return p;
  }
  Properties body(char end) {
    Properties p;
    for (;;) {
      ws();
      if (i >= s.size()) { if (end) fail("missing `}'"); break; }
      if (end && s[i] == end) { i++; break; }
      std::string key = ident(); ws();
      if (i >= s.size() || s[i] != '=') fail("`=' expected after " + key);
      i++;
      p.setAny(key, value()); ws();
      if (i < s.size() && s[i] == ';') i++;
    }
    return p;
  }
};

Properties Properties::parse(const std::string& text) { Parser ps(text); return ps.body(0); }

// =============================================================================================
//   Mesh
// =============================================================================================
Mesh Mesh::parse(const std::string& text) {
  Mesh m;
  std::map<int64_t, int32_t> nodeIndex, elemIndex;
  size_t i = 0;
  auto section = [&](std::string& tag, std::string& name, std::string& body) -> bool {
    size_t a = text.find('<', i);
    while (a != std::string::npos && (text.compare(a, 4, "<!--") == 0)) { a = text.find("-->", a); if (a == std::string::npos) return false; a = text.find('<', a); }
    if (a == std::string::npos) return false;
    size_t b = text.find('>', a);
    std::string head = text.substr(a + 1, b - a - 1);
    std::istringstream hs(head); hs >> tag;
    name.clear();
    size_t q = head.find("name=");
    if (q != std::string::npos) { size_t q1 = head.find('"', q), q2 = head.find('"', q1 + 1); name = head.substr(q1 + 1, q2 - q1 - 1); }
    if (!head.empty() && head.back() == '/') { body.clear(); i = b + 1; return true; }
    size_t c = text.find("</" + tag + ">", b);
    if (c == std::string::npos) throw Error("Mesh::parse", "missing </" + tag + ">");
    body = text.substr(b + 1, c - b - 1);
    i = c + tag.size() + 3;
    return true;
  };
  std::string tag, name, body;
  while (section(tag, name, body)) {
    if (tag == "Nodes") {
      std::istringstream bs(body); std::string line;
      while (std::getline(bs, line, ';')) {
        std::istringstream ls(line); int64_t id; if (!(ls >> id)) continue;
        std::vector<double> x; double v; while (ls >> v) x.push_back(v);
        if (m.rank == 0) m.rank = (int)x.size();
        if ((int)x.size() != m.rank) throw Error("Mesh::parse", "node " + std::to_string(id) + " has a different rank");
        nodeIndex[id] = (int32_t)m.nodeIds.size(); m.nodeIds.push_back(id);
        m.coords.insert(m.coords.end(), x.begin(), x.end());
      }
    } else if (tag == "Elements") {
      std::istringstream bs(body); std::string line;
      while (std::getline(bs, line, ';')) {
        std::istringstream ls(line); int64_t id; if (!(ls >> id)) continue;
        std::vector<int32_t> nd; int64_t n;
        while (ls >> n) {
          auto it = nodeIndex.find(n);
          if (it == nodeIndex.end()) throw Error("Mesh::parse", "element " + std::to_string(id) + " refers to undefined node " + std::to_string(n));
          nd.push_back(it->second);
        }
        elemIndex[id] = (int32_t)m.elemIds.size(); m.elemIds.push_back(id); m.elems.push_back(nd);
      }
    } else if (tag == "ElementGroup" || tag == "NodeGroup") {
      std::vector<int32_t> idx; std::string t = body;
      for (char& ch : t) if (ch == '{' || ch == '}' || ch == ',') ch = ' ';
      std::istringstream bs(t); int64_t id;
      auto& index = (tag == "ElementGroup") ? elemIndex : nodeIndex;
      while (bs >> id) {
        auto it = index.find(id);
        if (it == index.end()) throw Error("Mesh::parse", tag + " " + name + " refers to undefined item " + std::to_string(id));
        idx.push_back(it->second);
      }
      ((tag == "ElementGroup") ? m.elemGroups : m.nodeGroups)[name] = idx;
    }
  }
  return m;
}

// =============================================================================================
//   XDofSpace / MatrixBuilder
// =============================================================================================
int XDofSpace::addType(const std::string& name) {
  for (size_t t = 0; t < names_.size(); t++) if (names_[t] == name) return (int)t;
  names_.push_back(name);
  return (int)names_.size() - 1;
}
void XDofSpace::addDofs(const std::vector<int32_t>& inodes, const std::vector<int>& types) {
  for (int32_t n : inodes) for (int t : types) {
    auto key = std::make_pair(n, t);
    if (!index_.count(key)) index_[key] = count_++;
  }
}
int64_t XDofSpace::getDofIndex(int32_t inode, int type) const {
  auto it = index_.find(std::make_pair(inode, type));
  return it == index_.end() ? -1 : it->second;
}

void MatrixBuilder::addBlock(const std::vector<int64_t>& irows, const std::vector<int64_t>& jcols, const double* block) {
  for (size_t i = 0; i < irows.size(); i++)
    for (size_t j = 0; j < jcols.size(); j++) {
      const int64_t lo = rowOffsets[irows[i]], hi = rowOffsets[irows[i] + 1];
      auto it = std::lower_bound(colIndices.begin() + lo, colIndices.begin() + hi, (int32_t)jcols[j]);
      if (it == colIndices.begin() + hi || *it != jcols[j]) throw Error("MatrixBuilder::addBlock", "entry outside the matrix structure");
      values[it - colIndices.begin()] += block[i * jcols.size() + j];
    }
}
double MatrixBuilder::getValue(int64_t i, int64_t j) const {
  const int64_t lo = rowOffsets[i], hi = rowOffsets[i + 1];
  auto it = std::lower_bound(colIndices.begin() + lo, colIndices.begin() + hi, (int32_t)j);
  return (it == colIndices.begin() + hi || *it != j) ? 0.0 : values[it - colIndices.begin()];
}

std::shared_ptr<MatrixBuilder> newFEMMatrixBuilder(const Mesh& mesh, const std::vector<int32_t>& ielems, const XDofSpace& dofs) {
  const int64_t ndof = dofs.dofCount();
  std::vector<std::set<int32_t>> rows(ndof);
  std::vector<int64_t> ed;
  for (int32_t ie : ielems) {
    ed.clear();
    for (int32_t n : mesh.elems[ie]) for (int t = 0; t < dofs.typeCount(); t++) { int64_t d = dofs.getDofIndex(n, t); if (d >= 0) ed.push_back(d); }
    for (int64_t r : ed) for (int64_t c : ed) rows[r].insert((int32_t)c);
  }
  auto mb = std::make_shared<MatrixBuilder>();
  mb->rowOffsets.assign(ndof + 1, 0);
  for (int64_t r = 0; r < ndof; r++) mb->rowOffsets[r + 1] = mb->rowOffsets[r] + (int64_t)rows[r].size();
  mb->colIndices.reserve(mb->rowOffsets[ndof]);
  for (int64_t r = 0; r < ndof; r++) mb->colIndices.insert(mb->colIndices.end(), rows[r].begin(), rows[r].end());
  mb->values.assign(mb->colIndices.size(), 0.0);
  return mb;
}

// =============================================================================================
//   ModelFactory
// =============================================================================================
static std::map<std::string, ModelCtor>& registry() { static std::map<std::string, ModelCtor> r; return r; }
void ModelFactory::declare(const std::string& type, ModelCtor ctor) { registry()[type] = ctor; }
bool ModelFactory::exists(const std::string& type) { return registry().count(type) != 0; }
std::shared_ptr<Model> ModelFactory::newInstance(const std::string& name, Properties& conf, const Properties& props, Properties& globdat) {
  std::string type;
  props.getProps(name).get(type, "type");
  auto it = registry().find(type);
  if (it == registry().end()) throw Error(name, "unknown model type: " + type);
  return it->second(name, conf, props, globdat);
}

// =============================================================================================
//   The four models of the hot path
// =============================================================================================
namespace {

int shapeId(const std::string& type, const std::string& scheme, const std::string& ctx) {
  struct S { const char* t; const char* s; int id; };
  static const S tab[] = { { "Triangle3", "Gauss1", FA_TRI3_G1 }, { "Triangle3", "Gauss3", FA_TRI3_G3 },
    { "Quad4", "Gauss2*Gauss2", FA_QUAD4_G2 }, { "Quad8", "Gauss3*Gauss3", FA_QUAD8_G3 }, { "Quad8", "Gauss2*Gauss2", FA_QUAD8_G2 },
    { "Triangle6", "Gauss3", FA_TRI6_G3 }, { "Hex8", "Gauss2*Gauss2*Gauss2", FA_HEX8_G2 },
    { "Hex20", "Gauss3*Gauss3*Gauss3", FA_HEX20_G3 }, { "Hex20", "Gauss2*Gauss2*Gauss2", FA_HEX20_G2 } };
  for (auto& e : tab) if (type == e.t && scheme == e.s) return e.id;
  throw Error(ctx, "unsupported shape / integration scheme: " + type + " / " + scheme);
}
int shapeRank(int id) { return id >= FA_HEX8_G2 ? 3 : 2; }
int shapeNodes(int id) { static const int n[] = { 3, 3, 4, 8, 8, 6, 8, 20, 20 }; return n[id]; }

class SolidAssemblyModel : public Model {
 public:
  SolidAssemblyModel(int kind, const std::string& name, Properties& conf, const Properties& props, Properties& globdat)
      : Model(name), kind_(kind) {
    Properties myProps = props.getProps(myName_);
    Properties myConf = conf.makeProps(myName_);
    const std::string context = "model `" + myName_ + "'";
    // ElemGroup::get: the "elements" property names the group (PhaseFractureModel.cpp:102)
    std::string gname; myProps.get(gname, "elements"); myConf.set("elements", gname);
    globdat.get(mesh_, Globdat::MESH);
    auto git = mesh_->elemGroups.find(gname);
    if (git == mesh_->elemGroups.end()) throw Error(context, "undefined element group: " + gname);
    ielems_ = git->second;
    rank_ = mesh_->rank;
    if (rank_ < 1 || rank_ > 3) throw Error(context, "invalid node rank: " + std::to_string(rank_) + " (should be 1, 2 or 3)");
    Properties shProps = myProps.getProps("shape");
    std::string stype, scheme; shProps.get(stype, "type"); shProps.get(scheme, "intScheme");
    shape_ = shapeId(stype, scheme, context);
    if (shapeRank(shape_) != rank_)
      throw Error(context, "shape has invalid rank: " + std::to_string(shapeRank(shape_)) + " (should be " + std::to_string(rank_) + ")");
    for (int32_t ie : ielems_)                       // elems_.checkSomeElements
      if ((int)mesh_->elems[ie].size() != shapeNodes(shape_))
        throw Error(context, "element " + std::to_string(mesh_->elemIds[ie]) + " has " + std::to_string(mesh_->elems[ie].size()) +
                             " nodes (should be " + std::to_string(shapeNodes(shape_)) + ")");
    // DOF space (PhaseFractureModel.cpp:157-179)
    if (!globdat.find(dofs_, Globdat::DOFS)) { dofs_ = std::make_shared<XDofSpace>(); globdat.set(Globdat::DOFS, dofs_); }
    static const char* DISP[3] = { "dx", "dy", "dz" };
    for (int i = 0; i < rank_; i++) types_.push_back(dofs_->addType(DISP[i]));
    // the last DOF type: the phase field, or the nonlocal equivalent strain (GradientDamageModel.cpp:156-158)
    if (kind_ != FA_LINEAR_ELASTICITY) types_.push_back(dofs_->addType(kind_ == FA_GRADIENT_DAMAGE ? "eqv" : "phi"));
    std::set<int32_t> uniq;
    for (int32_t ie : ielems_) uniq.insert(mesh_->elems[ie].begin(), mesh_->elems[ie].end());
    nodes_.assign(uniq.begin(), uniq.end());
    dofs_->addDofs(nodes_, types_);
    // newMaterial (Material.cpp:212-247)
    Properties matProps = myProps.getProps("material");
    Properties matConf = myConf.makeProps("material");
    std::string mtype; int mrank = 0;
    matProps.get(mtype, "type"); matConf.set("type", mtype);
    matProps.get(mrank, "rank"); matConf.set("rank", mrank);
    if (mtype == "Hooke") material_ = FA_HOOKE;
    else if (mtype == "AmorPhase") material_ = FA_AMOR_PHASE;
    else if (mtype == "BourdinPhase") material_ = FA_BOURDIN_PHASE;
    else if (mtype == "MiehePhase") material_ = FA_MIEHE_PHASE;
    else if (mtype == "Damage") material_ = FA_DAMAGE;
    else throw Error(matProps.context, "Invalid material: " + mtype);
    if (mrank < 1 || mrank > 3) throw Error(matProps.context, "invalid material rank");
    if (kind_ == FA_GRADIENT_DAMAGE) {
      if (material_ != FA_DAMAGE) throw Error(context, "material type is not a damage material");   // GradientDamageModel.cpp:181-189
    } else if (kind_ != FA_LINEAR_ELASTICITY) {
      if (material_ == FA_HOOKE || material_ == FA_DAMAGE) throw Error(context, "material is not a PhaseFractureMaterial");   // dynamicCast, :194
      myProps.find(keepOffDiags_, "keepOffDiags"); myConf.set("keepOffDiags", keepOffDiags_);
      std::string ft; myProps.get(ft, "fractureType");
      static const char* FT[6] = { "at1", "at2", "czmLinear", "czmExponential", "czmCornelissen", "czmHyperbolic" };
      fracType_ = -1;
      for (int k = 0; k < 6; k++) if (ft == FT[k]) fracType_ = k;
      if (fracType_ < 0)
        throw Error(context, "unexpected definition of fracture type!!!\nSupported types include: \n"
                             "at1, at2, czmLinear, czmExponential, czmCornelissen, czmHyperbolic");
      myConf.set("fractureType", ft);
      myProps.find(gc_, "griffithEnergy"); myConf.set("griffithEnergy", gc_);
      myProps.find(l0_, "lengthScale"); myConf.set("lengthScale", l0_);
      myProps.find(sigmaT_, "tensileStrength"); myConf.set("tensileStrength", sigmaT_);
      if (kind_ == FA_MICRO_PHASE_FRACTURE) beta_ = 250.0;               // MicroPhaseFractureModel.cpp:269
      if (kind_ >= FA_PHASE_FRACTURE_EXT) { myProps.find(beta_, "penalty"); myConf.set("penalty", beta_); }
      if (kind_ == FA_MICRO_PHASE_FRACTURE_EXTIT) {                      // MicroPhaseFractureExtItModel.cpp:262-268
        oTol_ = 1.e-3; myProps.find(oTol_, "oTol"); myConf.set("oTol", oTol_);
        double mi = oMaxIter_; if (myProps.find(mi, "oMaxIter")) oMaxIter_ = (int)mi; myConf.set("oMaxIter", (double)oMaxIter_);
      }
    }
  }
  ~SolidAssemblyModel() override { if (ctx_) fa_destroy(ctx_); }

  void configure(const Properties& props, const Properties& globdat) override {
    Properties matProps = props.findProps(myName_).findProps("material");
    double v;
    if (matProps.find(v, "young")) { if (v < 0) throw Error(matProps.context, "property `young' out of range"); young_ = v; }
    if (matProps.find(v, "poisson")) { if (v < 0 || v > 0.5) throw Error(matProps.context, "property `poisson' out of range"); poisson_ = v; }
    double K, G;
    if (matProps.find(K, "bulk_modulus") && matProps.find(G, "shear_modulus")) {
      young_ = 9 * K * G / (3 * K + G);
      poisson_ = (3 * K - 2 * G) / 2. / (3 * K + G);
    }
    if (rank_ == 2) {
      std::string st; matProps.get(st, "state");
      if (st == "PLANE_STRAIN") state_ = FA_PLANE_STRAIN;
      else if (st == "PLANE_STRESS") state_ = FA_PLANE_STRESS;
      else state_ = -1;                                     // AXISYMMETRIC etc.: HookeMaterial.cpp:262-265 throws later
    }
    if (material_ == FA_DAMAGE) configureDamage_(matProps);
    std::shared_ptr<MatrixBuilder> mb;
    globdat.find(mb, "matrix0");
    createContext_(mb.get());
  }

  // GradientDamageModel::setConstraints_ (GradientDamageModel.cpp:682-846): the nonlocal strain is interpolated one order
  // below the displacements, so the eqv dof of every mid-side node is tied to the mean of the edge's two corner dofs
  // (coefficients 0.5); a dof that is already a slave is left alone; linear shapes return at once; shapes other than
  // Line3 / Triangle6 / Quad8 are an IllegalInputException
  void setConstraints_(Constraints& cons) {
    if (shape_ == FA_TRI3_G1 || shape_ == FA_TRI3_G3 || shape_ == FA_QUAD4_G2) return;          // :705-724
    // :789-799 -- hexahedra, 8 nodes included, are in neither of the reference's lists (Jive's geometry name is not
    // visible here, so the message carries the node count only)
    if (shape_ == FA_HEX8_G2) throw Error(myName_, "unsupported shape geometry: hexahedron, 8 nodes");
    if (shape_ == FA_HEX20_G2 || shape_ == FA_HEX20_G3) throw Error(myName_, "unsupported shape geometry: hexahedron, 20 nodes");
    const int nedge = (shape_ == FA_TRI6_G3) ? 3 : 4, npe = 2 * nedge;                          // Jive order: corner, mid, corner, ...
    const int eqvType = types_.back();
    const std::vector<double> coeffs = { 0.5, 0.5 };
    for (int32_t ie : ielems_) {
      const std::vector<int32_t>& inodes = mesh_->elems[ie];
      for (int j = 0; j < nedge; j++) {                                                         // cdofs columns (:747-788)
        const int64_t idof = dofs_->getDofIndex(inodes[2 * j + 1], eqvType);
        if (cons.isSlaveDof(idof)) continue;
        cons.addConstraint(idof, { dofs_->getDofIndex(inodes[2 * j], eqvType), dofs_->getDofIndex(inodes[(2 * j + 2) % npe], eqvType) }, coeffs);
      }
    }
  }

  // DamageMaterial::configure (src/materials/DamageMaterial.cpp:157-315): same property names, ranges and messages
  void configureDamage_(const Properties& props) {
    if (!props.find(kappaI_, "kappaI")) {
      double ft = 0.0; props.get(ft, "ft");
      kappaI_ = ft / young_;
    }
    if (kappaI_ < 0) throw Error(props.context, "property `kappaI' out of range");
    props.find(lengthScale_, "lengthScale");
    std::string soft; props.get(soft, "softening");
    static const char* LAWS[11] = { "perfect", "linear1", "linear2", "exponential1", "exponential2", "power", "hyperbolic",
                                    "exponential2Reg", "expoEnergy", "polynomial", "multilinear" };   // softeningLawDefs, :134-144
    int law = -1;
    for (int k = 0; k < 11; k++) if (soft == LAWS[k]) law = k;
    if (law < 0)
      throw Error("DamageMaterial::configure", "unexpected definition of softening law!!!\nSupported softening laws include: \n"
                  "perfect, linear1, linear2, exponential1, exponential2, power, hyperbolic, polynomial, multilinear");
    softening_ = law;
    auto need = [&](double& v, const char* name, double lo) {           // props.get (v, name, lo, maxOf (v))
      props.get(v, name);
      if (v < lo) throw Error(props.context, std::string("property `") + name + "' out of range");
    };
    if (law == FA_SOFT_EXPONENTIAL1) { need(softAlpha_, "alpha", 0.0); need(softBeta_, "beta", 0.0); }                 // :210-216
    if (law == FA_SOFT_EXPONENTIAL2 || law == FA_SOFT_LINEAR1 || law == FA_SOFT_LINEAR2 || law == FA_SOFT_POLYNOMIAL)
      need(kappaC_, "kappaC", 0.0);                                                                                    // :217-222, :241-252, :261-266
    if (law == FA_SOFT_POWER) { need(kappaC_, "kappaC", 0.0); need(softAlpha_, "alpha", 0.0); need(softBeta_, "beta", 0.0); }   // :227-234
    if (law == FA_SOFT_HYPERBOLIC) need(softB_, "b", 0.0);                                                             // :235-240
    if (law == FA_SOFT_EXPOENERGY) { need(kappaI_, "kappaI", 0.0); need(kappaC_, "kappaC", 0.0); need(softS_, "s", 0.0); }      // :253-260
    if (law == FA_SOFT_MULTILINEAR) { need(kappaP_, "kappaP", 0.0); need(kappaC_, "kappaC", kappaP_); }                // :267-273
    std::string eq; props.get(eq, "eqvStrain");
    if (eq == "Mazars") eqvStrain_ = FA_EQV_MAZARS;
    else if (eq == "Energy") eqvStrain_ = FA_EQV_ENERGY;
    else if (eq == "vonMises") {
      if (rank_ != 3)
        throw Error(props.context, "eqvStrain = vonMises is not available on the GPU assembly path for 2-D models (its derivative is a "
                                   "6-vector for 2-D strains in the reference, DamageMaterial.cpp:677-683)");
      props.get(eta_, "eta");                                                                                          // :295-300
      if (eta_ < 0) throw Error(props.context, "property `eta' out of range");
      eqvStrain_ = FA_EQV_VONMISES;
    }
    else throw Error("DamageMaterial::configure", "unexpected definition of equivalent strain!!!\nAvailable ones are: Mazars, vonMises, Energy");
  }

  void getConfig(Properties& conf, const Properties&) const override {
    Properties matConf = conf.makeProps(myName_).makeProps("material");
    matConf.set("young", young_); matConf.set("poisson", poisson_); matConf.set("state", state_);
  }

  bool takeAction(const std::string& action, Properties& params, Properties& globdat) override {
    if (action == Actions::GET_INT_VECTOR) {
      if (kind_ == FA_LINEAR_ELASTICITY) return false;      // LinearElasticityModel.cpp:213-289 has no such branch
      Vector force; params.get(force, ActionParams::INT_VECTOR);
      getMatrix_(nullptr, force, globdat);
      return true;
    }
    if (action == Actions::GET_MATRIX0) {
      std::shared_ptr<MatrixBuilder> mb; Vector force;
      params.get(mb, ActionParams::MATRIX0); params.get(force, ActionParams::INT_VECTOR);
      getMatrix_(mb.get(), force, globdat);
      return true;
    }
    if (action == Actions::GET_MATRIX2) throw Error(myName_, "GET_MATRIX2 is not served by the GPU assembly path");
    if (kind_ == FA_GRADIENT_DAMAGE && action == "INIT") {                 // GradientDamageModel.cpp:321-330, setConstraints_ :682-724
      std::shared_ptr<Constraints> cons;                                 // Constraints::get ( dofs_, globdat )
      if (!globdat.find(cons, Globdat::CONSTRAINTS)) { cons = std::make_shared<Constraints>(); globdat.set(Globdat::CONSTRAINTS, cons); }
      setConstraints_(*cons);
      return true;
    }
    if (action == Actions::COMMIT) {
      if (kind_ == FA_PHASE_FRACTURE || kind_ >= FA_MICRO_PHASE_FRACTURE) check_(fa_commit(ctx_));   // newHist_.swap(preHist_)
      if (kind_ == FA_PHASE_FRACTURE_EXT) dt0_ = dt_;                    // PhaseFractureExtModel.cpp:450
      if (kind_ == FA_PHASE_FRACTURE_EXTIT) oIter_ = 0;                  // PhaseFractureExtItModel.cpp:480
      return true;
    }
    if (kind_ == FA_PHASE_FRACTURE_EXTIT || kind_ == FA_MICRO_PHASE_FRACTURE_EXTIT) {
      if (action == "checkCommit") { checkCommit_(params, globdat); return true; }   // XActions::CHECK_COMMIT
    } else if (kind_ == FA_MICRO_PHASE_FRACTURE_EXT) {
      if (action == "checkCommit") return true;            // empty checkCommit_, MicroPhaseFractureExtModel.cpp:494-498
    } else if (action == Actions::CHECK_COMMIT) return true;
    if (action == Actions::SET_STEP_SIZE && (kind_ == FA_PHASE_FRACTURE_EXT || kind_ == FA_PHASE_FRACTURE_EXTIT)) {
      double dt, dt0; params.get(dt, ActionParams::STEP_SIZE); params.get(dt0, ActionParams::STEP_SIZE_0);
      dt_ = dt;
      if (kind_ == FA_PHASE_FRACTURE_EXTIT) dt0_ = dt0;                  // Ext ignores dt0 (PhaseFractureExtModel.cpp:1339)
      return false;                                                      // sic: both reference models fall through
    }
    if (action == Actions::SET_STEP_SIZE && (kind_ == FA_MICRO_PHASE_FRACTURE_EXT || kind_ == FA_MICRO_PHASE_FRACTURE_EXTIT)) {   // MicroPhaseFractureExtModel.cpp:1560-1577
      params.get(dt_, ActionParams::STEP_SIZE); params.get(dt0_, ActionParams::STEP_SIZE_0);
      return false;
    }
    return false;
  }

  fa_ctx* context() const { return ctx_; }
  double errExt() const { return errExt_; }

 private:
  void check_(fa_status st) const { if (st != FA_OK) throw Error(myName_, fa_last_error()); }

  void createContext_(const MatrixBuilder* mb) {
    const int nd = (int)types_.size(), npe = shapeNodes(shape_);
    std::map<int32_t, int32_t> local;
    for (size_t k = 0; k < nodes_.size(); k++) local[nodes_[k]] = (int32_t)k;
    std::vector<double> xc(nodes_.size() * rank_);
    for (size_t k = 0; k < nodes_.size(); k++)
      for (int i = 0; i < rank_; i++) xc[k * rank_ + i] = mesh_->coords[(size_t)nodes_[k] * rank_ + i];
    std::vector<int32_t> conn; conn.reserve(ielems_.size() * npe);
    for (int32_t ie : ielems_) for (int32_t n : mesh_->elems[ie]) conn.push_back(local[n]);
    std::vector<int64_t> nodeDofs(nodes_.size() * nd);
    bool canonical = (dofs_->dofCount() == (int64_t)nodes_.size() * nd);
    for (size_t k = 0; k < nodes_.size(); k++)
      for (int t = 0; t < nd; t++) {
        nodeDofs[k * nd + t] = dofs_->getDofIndex(nodes_[k], types_[t]);
        canonical = canonical && nodeDofs[k * nd + t] == (int64_t)(k * nd + t);
      }
    fa_problem p{};
    p.shape = shape_; p.model = kind_; p.material = material_; p.state = state_;
    p.young = young_; p.poisson = poisson_; p.frac_type = fracType_ < 0 ? FA_AT2 : fracType_;
    p.gc = gc_; p.l0 = l0_; p.sigma_t = sigmaT_; p.beta = beta_; p.keep_off_diags = keepOffDiags_ ? 1 : 0;
    p.softening = softening_; p.kappa_i = kappaI_; p.kappa_c = kappaC_; p.soft_alpha = softAlpha_; p.soft_beta = softBeta_;
    p.soft_b = softB_; p.soft_s = softS_; p.kappa_p = kappaP_; p.eta = eta_;
    p.length_scale = lengthScale_; p.eqv_strain = eqvStrain_;
    p.nnode = (int64_t)nodes_.size(); p.coords = xc.data();
    p.nel = (int64_t)ielems_.size(); p.conn = conn.data();
    p.ndof = dofs_->dofCount();
    p.node_dofs = canonical ? nullptr : nodeDofs.data();
    if (mb) { p.row_ptr = mb->rowOffsets.data(); p.col_idx = mb->colIndices.data(); }
    else if (!canonical) throw Error(myName_, "a shared DOF space needs the matrix builder (globdat `matrix0') at configure time");
    p.pattern_kind = FA_PATTERN_FULL; p.device = 0;
    if (ctx_) { fa_destroy(ctx_); ctx_ = nullptr; }
    check_(fa_create(&p, &ctx_));
  }

  // getMatrix_ (PhaseFractureModel.cpp:502-751 and variants): one call into the HIP library
  void getMatrix_(MatrixBuilder* mb, const Vector& force, Properties& globdat) {
    Vector state, state0, state00;
    globdat.get(state, Globdat::STATE);                                  // StateVector::get
    const double* s0 = nullptr; const double* s00 = nullptr;
    if (kind_ == FA_PHASE_FRACTURE_EXT || kind_ == FA_PHASE_FRACTURE_EXTIT) {
      globdat.get(state0, Globdat::STATE_OLD); globdat.get(state00, Globdat::STATE_OLD_OLD);
      s0 = state0->data(); s00 = state00->data();
      if (kind_ == FA_PHASE_FRACTURE_EXTIT) {                            // PhaseFractureExtItModel.cpp:399-408
        int iterCount = 0; globdat.find(iterCount, Globdat::ITER_COUNT);
        if (iterCount == 0 && !extFail_) {
          stateExt_.resize(state->size());
          const double c = (dt_ + dt0_) / (dt0_ + 1.e-15);
          for (size_t k = 0; k < state->size(); k++) stateExt_[k] = (*state00)[k] + c * ((*state0)[k] - (*state00)[k]);
        }
        s00 = stateExt_.data();
      }
    }
    const bool microExt = kind_ == FA_MICRO_PHASE_FRACTURE_EXT || kind_ == FA_MICRO_PHASE_FRACTURE_EXTIT;
    if (microExt) {                                                      // MicroPhaseFractureExtModel.cpp:398-417, ExtIt :413-432
      globdat.get(state0, Globdat::STATE_OLD); globdat.get(state00, Globdat::STATE_OLD_OLD);
      int iterCount = 0; globdat.find(iterCount, Globdat::ITER_COUNT);
      if (iterCount == 0 && !(kind_ == FA_MICRO_PHASE_FRACTURE_EXTIT && extFail_)) {
        stateExt_.resize(state->size());
        const double c = (dt_ + dt0_) / (dt0_ + 1.e-15);
        for (size_t k = 0; k < state->size(); k++) stateExt_[k] = (*state00)[k] + c * ((*state0)[k] - (*state00)[k]);
      }
      if (stateExt_.size() != state->size()) throw Error(myName_, "stateExt_ has not been built (no call at iteration 0 yet)");
      s00 = stateExt_.data();
    }
    if ((int64_t)state->size() != fa_ndof(ctx_) || (int64_t)force->size() != fa_ndof(ctx_))
      throw Error(myName_, "state / force vector size does not match the DOF space");
    if (mb && (int64_t)mb->values.size() != fa_nnz(ctx_)) throw Error(myName_, "matrix structure differs from the one given at configure time");
    double err = 0.0;
    check_(fa_assemble_host(ctx_, state->data(), s0, s00, dt_, dt0_, mb ? mb->values.data() : nullptr, force->data(),
                            (kind_ == FA_PHASE_FRACTURE_EXTIT || kind_ == FA_MICRO_PHASE_FRACTURE_EXTIT) ? &err : nullptr, FA_ACCUMULATE));
    if (kind_ == FA_PHASE_FRACTURE_EXTIT || kind_ == FA_MICRO_PHASE_FRACTURE_EXTIT) errExt_ = err;
  }

  void checkCommit_(Properties& params, Properties& globdat) {          // PhaseFractureExtItModel.cpp:1335-1373
    Vector state; globdat.get(state, Globdat::STATE);
    if (kind_ == FA_MICRO_PHASE_FRACTURE_EXTIT) {                        // MicroPhaseFractureExtItModel.cpp:1575-1613
      if (std::sqrt(errExt_) > oTol_ && oIter_ < oMaxIter_) {
        oIter_++;
        params.set("discard", false); params.set("accept", false);
        stateExt_ = *state;
        extFail_ = true;
      } else { extFail_ = false; oIter_ = 0; }
      errExt_ = 0.0;
      return;
    }
    if (std::sqrt(errExt_) > 1.e-4) {
      oIter_++;
      params.set("discard", false); params.set("accept", false);
      for (size_t k = 0; k < stateExt_.size(); k++) stateExt_[k] = 0.25 * stateExt_[k] + 0.75 * (*state)[k];
      extFail_ = true;
    } else {
      extFail_ = false;
    }
    errExt_ = 0.0;
  }

  int kind_;
  std::shared_ptr<Mesh> mesh_; std::shared_ptr<XDofSpace> dofs_;
  std::vector<int32_t> ielems_, nodes_; std::vector<int> types_;
  int rank_ = 0, shape_ = 0, material_ = 0, state_ = FA_PLANE_STRAIN, fracType_ = -1;
  double young_ = 1.0, poisson_ = 0.0;                        // HookeMaterial.cpp:44-45 defaults
  bool keepOffDiags_ = false;
  double gc_ = 2.7, l0_ = 0.015, sigmaT_ = 0.0, beta_ = 6500.0;   // PhaseFractureModel.cpp:253-263, Ext :247
  double dt_ = 0.0, dt0_ = 0.0, errExt_ = 0.0;
  int softening_ = 0, eqvStrain_ = 0;                         // DamageMaterial (GradientDamage)
  double kappaI_ = 0.0, kappaC_ = 0.0, softAlpha_ = 0.0, softBeta_ = 0.0, lengthScale_ = 0.0, softB_ = 0.0, softS_ = 0.0, kappaP_ = 0.0, eta_ = 0.0;
  bool extFail_ = false; int oIter_ = 0, oMaxIter_ = 5000; double oTol_ = 1.e-3;
  std::vector<double> stateExt_;
  fa_ctx* ctx_ = nullptr;
};

template <int KIND>
std::shared_ptr<Model> newSolidModel(const std::string& name, Properties& conf, const Properties& props, Properties& globdat) {
  return std::make_shared<SolidAssemblyModel>(KIND, name, conf, props, globdat);
}

}  // namespace

void declareFalconSolidMechModels() {
  ModelFactory::declare("LinearElasticity", &newSolidModel<FA_LINEAR_ELASTICITY>);
  ModelFactory::declare("PhaseFracture", &newSolidModel<FA_PHASE_FRACTURE>);
  ModelFactory::declare("PhaseFractureExt", &newSolidModel<FA_PHASE_FRACTURE_EXT>);
  ModelFactory::declare("PhaseFractureExtIt", &newSolidModel<FA_PHASE_FRACTURE_EXTIT>);
  ModelFactory::declare("MicroPhaseFracture", &newSolidModel<FA_MICRO_PHASE_FRACTURE>);   // micromorphic variants
  ModelFactory::declare("MicroPhaseFractureExt", &newSolidModel<FA_MICRO_PHASE_FRACTURE_EXT>);
  ModelFactory::declare("MicroPhaseFractureExtIt", &newSolidModel<FA_MICRO_PHASE_FRACTURE_EXTIT>);
  ModelFactory::declare("GradientDamage", &newSolidModel<FA_GRADIENT_DAMAGE>);   // GradientDamageModel.cpp:1455-1459
}

}  // namespace falcon_host

// =============================================================================================
//   C entry points for the Python tests / demos (ctypes): build the model tree of a .pro + .mesh pair and
//   drive it through takeAction exactly like Jive's MatrixModel / solver would.
// =============================================================================================
using namespace falcon_host;

struct fh_case {
  Properties props, conf, globdat;
  std::shared_ptr<Mesh> mesh;
  std::shared_ptr<Model> model;
  std::shared_ptr<MatrixBuilder> matrix;
  std::string modelName, modelType;
};

static thread_local std::string g_fh_err;
extern "C" const char* fh_last_error() { return g_fh_err.c_str(); }

// find the first child of the model tree ("Matrix" -> "Multi" -> models[...]) whose type is on the hot path
static bool findBulk(const Properties& tree, std::string& path, Properties& parent, std::string& name) {
  std::string type;
  if (tree.find(type, "type")) {
    if (type == "Matrix" || type == "Multi") {
      std::vector<std::string> names;
      if (tree.find(names, "models")) {
        for (auto& n : names) {
          Properties child = tree.findProps(n);
          std::string ct;
          if (child.find(ct, "type") && ModelFactory::exists(ct)) { parent = tree; name = n; return true; }
          if (findBulk(child, path, parent, name)) return true;
        }
      }
      Properties inner = tree.findProps("model");
      return findBulk(inner, path, parent, name);
    }
  }
  return false;
}

extern "C" fh_case* fh_create(const char* pro_text, const char* mesh_text) {
  try {
    declareFalconSolidMechModels();
    auto c = std::make_unique<fh_case>();
    c->props = Properties::parse(pro_text);
    c->mesh = std::make_shared<Mesh>(Mesh::parse(mesh_text));
    c->globdat.set(Globdat::MESH, c->mesh);
    Properties top = c->props.getProps("model"), parent;
    std::string path;
    if (!findBulk(top, path, parent, c->modelName)) throw Error("fh_create", "no LinearElasticity / PhaseFracture* model in the model tree");
    parent.getProps(c->modelName).get(c->modelType, "type");
    c->model = ModelFactory::newInstance(c->modelName, c->conf, parent, c->globdat);
    std::shared_ptr<XDofSpace> dofs; c->globdat.get(dofs, Globdat::DOFS);
    std::string gname; parent.getProps(c->modelName).get(gname, "elements");
    c->matrix = newFEMMatrixBuilder(*c->mesh, c->mesh->elemGroups.at(gname), *dofs);
    c->globdat.set("matrix0", c->matrix);
    c->model->configure(parent, c->globdat);
    return c.release();
  } catch (const std::exception& e) { g_fh_err = e.what(); return nullptr; }
}
extern "C" void fh_destroy(fh_case* c) { delete c; }
extern "C" int64_t fh_ndof(fh_case* c) { std::shared_ptr<XDofSpace> d; c->globdat.get(d, Globdat::DOFS); return d->dofCount(); }
extern "C" int64_t fh_nnz(fh_case* c) { return (int64_t)c->matrix->values.size(); }
extern "C" const char* fh_model_type(fh_case* c) { return c->modelType.c_str(); }
extern "C" void fh_pattern(fh_case* c, int64_t* rp, int32_t* ci) {
  std::copy(c->matrix->rowOffsets.begin(), c->matrix->rowOffsets.end(), rp);
  std::copy(c->matrix->colIndices.begin(), c->matrix->colIndices.end(), ci);
}
// constraints held in globdat (GradientDamage INIT): count, then slave / two-masters-and-coefficients per constraint
extern "C" int64_t fh_constraint_count(fh_case* c) {
  std::shared_ptr<Constraints> cons;
  return c->globdat.find(cons, Globdat::CONSTRAINTS) ? (int64_t)cons->all().size() : 0;
}
extern "C" int64_t fh_constraints(fh_case* c, int64_t* slaves, int64_t* master_ptr, int64_t* masters, double* coefs, int64_t cap) {
  std::shared_ptr<Constraints> cons;
  if (!c->globdat.find(cons, Globdat::CONSTRAINTS)) return 0;
  int64_t k = 0, m = 0;
  master_ptr[0] = 0;
  for (const auto& q : cons->all()) {
    if (m + (int64_t)q.masters.size() > cap) return -1;
    slaves[k] = q.slave;
    for (size_t i = 0; i < q.masters.size(); i++) { masters[m] = q.masters[i]; coefs[m] = q.coefs[i]; m++; }
    master_ptr[++k] = m;
  }
  return k;
}
extern "C" int64_t fh_dof_index(fh_case* c, int64_t node_id, const char* type) {
  std::shared_ptr<XDofSpace> d; c->globdat.get(d, Globdat::DOFS);
  for (size_t k = 0; k < c->mesh->nodeIds.size(); k++)
    if (c->mesh->nodeIds[k] == node_id)
      for (int t = 0; t < d->typeCount(); t++) if (d->typeName(t) == type) return d->getDofIndex((int32_t)k, t);
  return -1;
}
// returns 1 / 0 = takeAction's bool, -1 = exception.  K (nnz) and f (ndof) are zeroed first, like
// SolverRunData::updateMatrix does, then handed over as MATRIX0 / INT_VECTOR.
extern "C" int fh_take_action(fh_case* c, const char* action, const double* state, const double* state0,
                              const double* state00, double step, double step0, int iter, double* K, double* f) {
  try {
    const int64_t n = fh_ndof(c);
    auto vec = [&](const double* p) { auto v = newVector(n); if (p) std::copy(p, p + n, v->begin()); return v; };
    if (state) c->globdat.set(Globdat::STATE, vec(state));
    if (state0) c->globdat.set(Globdat::STATE_OLD, vec(state0));
    if (state00) c->globdat.set(Globdat::STATE_OLD_OLD, vec(state00));
    c->globdat.set(Globdat::ITER_COUNT, iter);
    Properties params;
    Vector force = newVector(n);
    c->matrix->setZero();
    params.set(ActionParams::MATRIX0, c->matrix);
    params.set(ActionParams::INT_VECTOR, force);
    params.set(ActionParams::STEP_SIZE, step); params.set(ActionParams::STEP_SIZE_0, step0);
    const bool r = c->model->takeAction(action, params, c->globdat);
    if (K) std::copy(c->matrix->values.begin(), c->matrix->values.end(), K);
    if (f) std::copy(force->begin(), force->end(), f);
    return r ? 1 : 0;
  } catch (const std::exception& e) { g_fh_err = e.what(); return -1; }
}
