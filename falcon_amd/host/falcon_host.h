// falcon_host.h -- host side ABOVE the C ABI: the reference's plugin interface for the assembly path, rebuilt
// without Jem/Jive so that it compiles in this image.
//
// The reference's models are jive::model::Model subclasses created by name through ModelFactory and driven only
// through  bool takeAction(action, params, globdat)  (src/fem/solidmech/PhaseFractureModel.h:138-159,
// PhaseFractureModel.cpp:384-494,1532-1554).  The classes below keep those names, constructor arguments, property
// names, action strings and error behaviour; the Jive objects they talk to (Properties, XDofSpace, MatrixBuilder,
// StateVector, ElementGroup) are small stand-ins with the same method names.  On a Jive box the real classes
// replace the stand-ins and the body of getMatrix_ stays what it is here: one call into libfalcon_assemble.so
// (see INTEGRATION.md).
#pragma once
#include <any>
#include <cstdint>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/falcon_assemble.h"

namespace falcon_host {

struct Error : std::runtime_error {            // jem::Error / IllegalInputException stand-in
  std::string where;
  Error(const std::string& w, const std::string& m) : std::runtime_error(w + ": " + m), where(w) {}
};

// ---- jem::util::Properties stand-in: nested, string-keyed; parses the .pro syntax -------------------------
class Properties {
 public:
  using Ptr = std::shared_ptr<Properties>;
  Properties() : d_(std::make_shared<Data>()) {}
  static Properties parse(const std::string& text);                 // "key = value; block = "Type" { ... };"

  bool contains(const std::string& key) const;
  template <class T> bool find(T& out, const std::string& key) const;   // false if absent; throws on type mismatch
  template <class T> void get(T& out, const std::string& key) const;    // throws if absent
  template <class T> void set(const std::string& key, const T& v) { setAny(key, std::any(v)); }
  Properties getProps(const std::string& key) const;                 // throws if absent
  Properties findProps(const std::string& key) const;                // empty if absent
  Properties makeProps(const std::string& key);                      // creates if absent
  std::vector<std::string> keys() const;
  std::string context;                                               // for error messages

 private:
  struct Data { std::map<std::string, std::any> m; };
  std::shared_ptr<Data> d_;                                          // shallow-copy semantics like jem Properties
  const std::any* lookup(const std::string& key) const;
  void setAny(const std::string& key, std::any v);
  friend struct Parser;
};

using Vector = std::shared_ptr<std::vector<double>>;                 // jem::Array<double,1>: shallow handle
inline Vector newVector(size_t n, double v = 0.0) { return std::make_shared<std::vector<double>>(n, v); }

// ---- mesh (jive::fem NodeSet / ElementSet / groups) read from the reference's .mesh format -----------------
struct Mesh {
  int rank = 0;
  std::vector<int64_t> nodeIds; std::vector<double> coords;           // nnode x rank
  std::vector<int64_t> elemIds; std::vector<std::vector<int32_t>> elems;   // node INDICES
  std::map<std::string, std::vector<int32_t>> elemGroups, nodeGroups;  // indices
  static Mesh parse(const std::string& text);                          // <Nodes>, <Elements>, <ElementGroup>, <NodeGroup>
  int64_t nodeCount() const { return (int64_t)nodeIds.size(); }
};

// ---- jive::util::XDofSpace stand-in -----------------------------------------------------------------------
class XDofSpace {
 public:
  int addType(const std::string& name);                                // returns the type index, idempotent
  void addDofs(const std::vector<int32_t>& inodes, const std::vector<int>& types);   // node-major, type-minor
  int64_t dofCount() const { return count_; }
  int64_t getDofIndex(int32_t inode, int type) const;                  // -1 if absent
  int typeCount() const { return (int)names_.size(); }
  const std::string& typeName(int t) const { return names_[t]; }
 private:
  std::vector<std::string> names_;
  std::map<std::pair<int32_t, int>, int64_t> index_;
  int64_t count_ = 0;
};

// ---- jive::util::Constraints stand-in: linear constraints u[slave] = rvalue + sum coef * u[master], in insertion order ----
class Constraints {
 public:
  struct Con { int64_t slave; double rvalue; std::vector<int64_t> masters; std::vector<double> coefs; };
  bool isSlaveDof(int64_t idof) const { return slaveIndex_.count(idof) != 0; }
  void addConstraint(int64_t idof, const std::vector<int64_t>& jdofs, const std::vector<double>& coeffs) {
    if (isSlaveDof(idof)) { cons_[slaveIndex_[idof]] = Con{idof, 0.0, jdofs, coeffs}; return; }
    slaveIndex_[idof] = cons_.size(); cons_.push_back(Con{idof, 0.0, jdofs, coeffs});
  }
  const std::vector<Con>& all() const { return cons_; }
 private:
  std::vector<Con> cons_; std::map<int64_t, size_t> slaveIndex_;
};

// ---- jive::algebra::MatrixBuilder stand-in, matrix.type = "FEM": pattern fixed by mesh x dof space ----------
class MatrixBuilder {
 public:
  std::vector<int64_t> rowOffsets; std::vector<int32_t> colIndices; std::vector<double> values;
  void setZero() { std::fill(values.begin(), values.end(), 0.0); }
  void addBlock(const std::vector<int64_t>& irows, const std::vector<int64_t>& jcols, const double* block);
  double getValue(int64_t i, int64_t j) const;
};

struct Actions {                                                       // jive::model::Actions / ActionParams
  static constexpr const char* GET_MATRIX0 = "GET_MATRIX0";
  static constexpr const char* GET_INT_VECTOR = "GET_INT_VECTOR";
  static constexpr const char* GET_MATRIX2 = "GET_MATRIX2";
  static constexpr const char* COMMIT = "COMMIT";
  static constexpr const char* CHECK_COMMIT = "CHECK_COMMIT";
  static constexpr const char* GET_TABLE = "GET_TABLE";
  static constexpr const char* SET_STEP_SIZE = "setStepSize";          // XActions, src/util/XNames.cpp:52
};
struct ActionParams {
  static constexpr const char* MATRIX0 = "MATRIX0";
  static constexpr const char* INT_VECTOR = "INT_VECTOR";
  static constexpr const char* STEP_SIZE = "stepSize";                 // XProps, XNames.cpp:23-24
  static constexpr const char* STEP_SIZE_0 = "stepSize0";
};

// globdat keys used here
struct Globdat {
  static constexpr const char* MESH = "mesh";             // shared_ptr<Mesh>
  static constexpr const char* DOFS = "dofs";             // shared_ptr<XDofSpace>
  static constexpr const char* CONSTRAINTS = "constraints";   // shared_ptr<Constraints> (Constraints::get (dofs, globdat))
  static constexpr const char* STATE = "state";           // Vector (StateVector::get)
  static constexpr const char* STATE_OLD = "state0";      // StateVector::getOld
  static constexpr const char* STATE_OLD_OLD = "state00"; // StateVector::getOldOld
  static constexpr const char* ITER_COUNT = "solverInfo.iterCount";
};

class Model {                                                          // jive::model::Model
 public:
  explicit Model(const std::string& name) : myName_(name) {}
  virtual ~Model() = default;
  virtual void configure(const Properties& props, const Properties& globdat) {}
  virtual void getConfig(Properties& conf, const Properties& globdat) const {}
  virtual bool takeAction(const std::string& action, Properties& params, Properties& globdat) = 0;
  const std::string& getName() const { return myName_; }
 protected:
  std::string myName_;
};

using ModelCtor = std::shared_ptr<Model> (*)(const std::string& name, Properties& conf, const Properties& props,
                                             Properties& globdat);
struct ModelFactory {
  static void declare(const std::string& type, ModelCtor ctor);
  static bool exists(const std::string& type);
  static std::shared_ptr<Model> newInstance(const std::string& name, Properties& conf, const Properties& props,
                                            Properties& globdat);     // type = props[name].type
};
void declareFalconSolidMechModels();    // FalconSolidMechModels.cpp:10-22 (the four models on the hot path)

// Builds the "FEM" matrix for a model's DOF space (node pairs sharing an element x all dof types present)
std::shared_ptr<MatrixBuilder> newFEMMatrixBuilder(const Mesh& mesh, const std::vector<int32_t>& ielems,
                                                   const XDofSpace& dofs);

}  // namespace falcon_host
