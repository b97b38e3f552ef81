// duckdb.hpp — a STAND-IN for the parts of DuckDB's C++ API that duckdb_shim/fit_agg_hip.cpp touches (SURVEY.md Appendix E
// lists them as the reference uses them: src/aggregate_functions/ols_aggregate.cpp:3-12,103-426).  Test infrastructure:
// DuckDB's headers are not available in this repository (the reference's `duckdb` submodule is empty), so the glue is
// compiled — with -Wall -Wextra under ASan / UBSan — and driven against these declarations, which keep DuckDB's names,
// signatures and calling conventions (state vectors of pointers, unified formats with selection vectors, LIST / STRUCT
// vectors, bind data reached through AggregateInputData, function sets registered through an ExtensionLoader).
// Nothing here is shipped, and nothing here is DuckDB code.
#pragma once
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace duckdb {

using idx_t = uint64_t;
using data_t = uint8_t;
using data_ptr_t = data_t *;
using std::string;
using std::pair;
using std::make_pair;
using std::unique_ptr;
using std::shared_ptr;
using std::make_shared;
template <class T>
using vector = std::vector<T>;
template <class T, class... A>
unique_ptr<T> make_uniq(A &&...a) { return unique_ptr<T>(new T(std::forward<A>(a)...)); }
template <class T, class... A>
shared_ptr<T> make_shared_ptr(A &&...a) { return std::make_shared<T>(std::forward<A>(a)...); }
template <class T>
using child_list_t = vector<pair<string, T>>;
constexpr idx_t STANDARD_VECTOR_SIZE = 2048;

template <class T>
class optional_ptr {
public:
	optional_ptr() : p_(nullptr) {}
	optional_ptr(T *p) : p_(p) {} // NOLINT
	T *operator->() const { if (!p_) throw std::runtime_error("optional_ptr: dereferencing NULL"); return p_; }
	T &operator*() const { return *operator->(); }
	explicit operator bool() const { return p_ != nullptr; }
	T *get() const { return p_; }
private:
	T *p_;
};

class Exception : public std::runtime_error {
public:
	explicit Exception(const string &m) : std::runtime_error(m) {}
};
inline string FormatMessage(const char *fmt, va_list ap) {
	char buf[1024];
	vsnprintf(buf, sizeof buf, fmt, ap);
	return buf;
}
class InvalidInputException : public Exception {
public:
	explicit InvalidInputException(const string &m) : Exception(m) {}
	InvalidInputException(const char *fmt, ...) __attribute__((format(printf, 2, 3))) : Exception(Make(fmt)) {
		va_list ap;
		va_start(ap, fmt);
		msg_ = FormatMessage(fmt, ap);
		va_end(ap);
	}
	const char *what() const noexcept override { return msg_.empty() ? Exception::what() : msg_.c_str(); }
private:
	static string Make(const char *fmt) { return fmt; }
	string msg_;
};
class BinderException : public Exception {
public:
	explicit BinderException(const string &m) : Exception(m) {}
};

enum class LogicalTypeId : uint8_t { INVALID, ANY, BOOLEAN, INTEGER, BIGINT, DOUBLE, VARCHAR, LIST, STRUCT, MAP, POINTER };

class LogicalType {
public:
	LogicalType() : id_(LogicalTypeId::INVALID) {}
	LogicalType(LogicalTypeId id) : id_(id) {} // NOLINT
	LogicalTypeId id() const { return id_; }
	bool operator==(const LogicalType &o) const {
		if (id_ != o.id_ || children_.size() != o.children_.size()) return false;
		for (size_t i = 0; i < children_.size(); ++i)
			if (children_[i].first != o.children_[i].first || !(children_[i].second == o.children_[i].second)) return false;
		return true;
	}
	bool operator!=(const LogicalType &o) const { return !(*this == o); }
	static LogicalType LIST(const LogicalType &child) {
		LogicalType t(LogicalTypeId::LIST);
		t.children_.push_back(make_pair(string("child"), child));
		return t;
	}
	static LogicalType STRUCT(child_list_t<LogicalType> children) {
		LogicalType t(LogicalTypeId::STRUCT);
		t.children_ = std::move(children);
		return t;
	}
	static LogicalType MAP(const LogicalType &key, const LogicalType &value) {
		LogicalType t(LogicalTypeId::MAP);
		t.children_.push_back(make_pair(string("key"), key));
		t.children_.push_back(make_pair(string("value"), value));
		return t;
	}
	const child_list_t<LogicalType> &children() const { return children_; }
	string ToString() const {
		switch (id_) {
		case LogicalTypeId::ANY: return "ANY";
		case LogicalTypeId::BOOLEAN: return "BOOLEAN";
		case LogicalTypeId::INTEGER: return "INTEGER";
		case LogicalTypeId::BIGINT: return "BIGINT";
		case LogicalTypeId::DOUBLE: return "DOUBLE";
		case LogicalTypeId::VARCHAR: return "VARCHAR";
		case LogicalTypeId::LIST: return children_[0].second.ToString() + "[]";
		case LogicalTypeId::STRUCT: {
			string s = "STRUCT(";
			for (size_t i = 0; i < children_.size(); ++i) s += (i ? ", " : "") + children_[i].first + " " + children_[i].second.ToString();
			return s + ")";
		}
		case LogicalTypeId::MAP: return "MAP(" + children_[0].second.ToString() + ", " + children_[1].second.ToString() + ")";
		case LogicalTypeId::POINTER: return "POINTER";
		default: return "INVALID";
		}
	}
	static const LogicalTypeId ANY = LogicalTypeId::ANY;
	static const LogicalTypeId BOOLEAN = LogicalTypeId::BOOLEAN;
	static const LogicalTypeId INTEGER = LogicalTypeId::INTEGER;
	static const LogicalTypeId BIGINT = LogicalTypeId::BIGINT;
	static const LogicalTypeId DOUBLE = LogicalTypeId::DOUBLE;
	static const LogicalTypeId VARCHAR = LogicalTypeId::VARCHAR;
	static const LogicalTypeId POINTER = LogicalTypeId::POINTER;
private:
	LogicalTypeId id_;
	child_list_t<LogicalType> children_;
};
struct StructType {
	static idx_t GetChildCount(const LogicalType &t) { return t.children().size(); }
	static const string &GetChildName(const LogicalType &t, idx_t i) { return t.children()[i].first; }
	static const LogicalType &GetChildType(const LogicalType &t, idx_t i) { return t.children()[i].second; }
};

// ---- constant values (the options argument after constant folding) ----
class Value {
public:
	Value() : null_(true) {}
	static Value BOOLEAN(bool v) { Value r(LogicalTypeId::BOOLEAN); r.i_ = v; return r; }
	static Value INTEGER(int32_t v) { Value r(LogicalTypeId::INTEGER); r.i_ = v; return r; }
	static Value BIGINT(int64_t v) { Value r(LogicalTypeId::BIGINT); r.i_ = v; return r; }
	static Value DOUBLE(double v) { Value r(LogicalTypeId::DOUBLE); r.d_ = v; return r; }
	explicit Value(const string &s) : type_(LogicalTypeId::VARCHAR), null_(false), s_(s) {}
	explicit Value(const LogicalType &null_of) : type_(null_of), null_(true) {}
	static Value STRUCT(child_list_t<Value> kids) {
		child_list_t<LogicalType> t;
		Value r;
		for (auto &k : kids) { t.push_back(make_pair(k.first, k.second.type())); r.kids_.push_back(k.second); }
		r.type_ = LogicalType::STRUCT(std::move(t));
		r.null_ = false;
		return r;
	}
	// a MAP value is a list of {key, value} structs, as in DuckDB
	static Value MAP(const LogicalType &key_type, const LogicalType &value_type, vector<Value> keys, vector<Value> values) {
		Value r;
		r.type_ = LogicalType::MAP(key_type, value_type);
		r.null_ = false;
		for (size_t i = 0; i < keys.size(); ++i) r.kids_.push_back(Value::STRUCT({{"key", keys[i]}, {"value", values[i]}}));
		return r;
	}
	const LogicalType &type() const { return type_; }
	bool IsNull() const { return null_; }
	template <class T>
	T GetValue() const;
	string ToString() const {
		if (null_) return "NULL";
		switch (type_.id()) {
		case LogicalTypeId::VARCHAR: return s_;
		case LogicalTypeId::DOUBLE: return std::to_string(d_);
		case LogicalTypeId::BOOLEAN: return i_ ? "true" : "false";
		default: return std::to_string(i_);
		}
	}
	const vector<Value> &kids() const { return kids_; }
private:
	explicit Value(LogicalTypeId id) : type_(id), null_(false) {}
	LogicalType type_;
	bool null_ = true;
	int64_t i_ = 0;
	double d_ = 0.0;
	string s_;
	vector<Value> kids_;
	friend struct BooleanValue;
	friend struct StringValue;
};
template <>
inline double Value::GetValue<double>() const {
	if (null_) throw InvalidInputException("GetValue on NULL");
	switch (type_.id()) {
	case LogicalTypeId::DOUBLE: return d_;
	case LogicalTypeId::BOOLEAN: case LogicalTypeId::INTEGER: case LogicalTypeId::BIGINT: return (double)i_;
	case LogicalTypeId::VARCHAR: {
		char *end = nullptr;
		const double v = strtod(s_.c_str(), &end);
		if (end == s_.c_str() || *end) throw InvalidInputException("Could not convert string '%s' to DOUBLE", s_.c_str());
		return v;
	}
	default: throw InvalidInputException("Unimplemented type for cast");
	}
}
template <>
inline int64_t Value::GetValue<int64_t>() const {
	if (null_) throw InvalidInputException("GetValue on NULL");
	if (type_.id() == LogicalTypeId::DOUBLE) return (int64_t)d_;
	return i_;
}
struct BooleanValue { static bool Get(const Value &v) { return v.i_ != 0; } };
struct StringValue { static const string &Get(const Value &v) { return v.s_; } };
struct StructValue { static const vector<Value> &GetChildren(const Value &v) { return v.kids(); } };
struct MapValue { static const vector<Value> &GetChildren(const Value &v) { return v.kids(); } };
struct ListValue { static const vector<Value> &GetChildren(const Value &v) { return v.kids(); } };

// ---- vectors ----
struct list_entry_t {
	uint64_t offset;
	uint64_t length;
};
// DuckDB's 16-byte string handle: up to 12 characters inline, longer ones behind a pointer the vector's string heap owns
// (duckdb/common/types/string_type.hpp).  The stand-in keeps the layout and the accessors the glue uses.
struct string_t {
	static constexpr idx_t INLINE_LENGTH = 12;
	string_t() { memset(&value_, 0, sizeof value_); }
	string_t(const char *data, uint32_t len) {
		memset(&value_, 0, sizeof value_);
		value_.inlined.length = len;
		if (len <= INLINE_LENGTH) {
			memcpy(value_.inlined.inlined, data, len);
		} else {
			memcpy(value_.pointer.prefix, data, 4);
			value_.pointer.ptr = data;
		}
	}
	bool IsInlined() const { return GetSize() <= INLINE_LENGTH; }
	const char *GetData() const { return IsInlined() ? value_.inlined.inlined : value_.pointer.ptr; }
	idx_t GetSize() const { return value_.inlined.length; }
	string GetString() const { return string(GetData(), GetSize()); }
private:
	union {
		struct {
			uint32_t length;
			char prefix[4];
			const char *ptr;
		} pointer;
		struct {
			uint32_t length;
			char inlined[12];
		} inlined;
	} value_;
};
static_assert(sizeof(string_t) == 16, "string_t is 16 bytes in DuckDB");
inline idx_t StubTypeWidth(LogicalTypeId id) {
	switch (id) {
	case LogicalTypeId::BOOLEAN: return 1;
	case LogicalTypeId::INTEGER: return 4;
	case LogicalTypeId::VARCHAR: return 16;
	case LogicalTypeId::LIST: return sizeof(list_entry_t);
	case LogicalTypeId::STRUCT: return 0;
	default: return 8;
	}
}
class ValidityMask {
public:
	bool RowIsValid(idx_t i) const { return i >= invalid_.size() || !invalid_[i]; }
	void SetInvalid(idx_t i) { if (i >= invalid_.size()) invalid_.resize(i + 1, false); invalid_[i] = true; }
	void SetValid(idx_t i) { if (i < invalid_.size()) invalid_[i] = false; }
	bool AllValid() const { for (bool b : invalid_) if (b) return false; return true; }
private:
	vector<bool> invalid_;
};
struct SelectionVector {
	const uint32_t *sel = nullptr; // nullptr = the identity
	idx_t get_index(idx_t i) const { return sel ? sel[i] : i; }
};
struct UnifiedVectorFormat {
	const SelectionVector *sel = nullptr;
	data_ptr_t data = nullptr;
	ValidityMask validity;
	SelectionVector owned_sel;
	template <class T>
	static const T *GetData(const UnifiedVectorFormat &f) { return reinterpret_cast<const T *>(f.data); }
};
enum class VectorType : uint8_t { FLAT_VECTOR, CONSTANT_VECTOR, DICTIONARY_VECTOR };

class Vector {
public:
	explicit Vector(const LogicalType &type, idx_t capacity = STANDARD_VECTOR_SIZE) : type_(type), capacity_(capacity) {
		switch (type.id()) {
		case LogicalTypeId::LIST:
			buffer_.assign(capacity * sizeof(list_entry_t), 0);
			child_.push_back(make_uniq<Vector>(type.children()[0].second, 0));
			break;
		case LogicalTypeId::STRUCT:
			for (auto &c : type.children()) child_.push_back(make_uniq<Vector>(c.second, capacity));
			break;
		default: buffer_.assign(capacity * StubTypeWidth(type.id()), 0); // exactly as wide as DuckDB's physical type: ASan sees an overrun
		}
	}
	// stand-in helper: a VARCHAR vector's strings live as long as the vector (DuckDB: its string heap)
	string_t AddString(const string &s) {
		heap_.push_back(make_uniq<string>(s));
		return string_t(heap_.back()->data(), (uint32_t)heap_.back()->size());
	}
	const LogicalType &GetType() const { return type_; }
	VectorType GetVectorType() const { return vtype_; }
	// stand-in helpers of the test driver: make this vector a constant / a dictionary over its flat data
	void MakeConstant() { vtype_ = VectorType::CONSTANT_VECTOR; }
	void MakeDictionary(vector<uint32_t> sel) { vtype_ = VectorType::DICTIONARY_VECTOR; dict_ = std::move(sel); }
	void ToUnifiedFormat(idx_t count, UnifiedVectorFormat &f) {
		f.data = buffer_.data();
		f.validity = validity_;
		if (vtype_ == VectorType::FLAT_VECTOR) {
			f.owned_sel.sel = nullptr;
		} else if (vtype_ == VectorType::CONSTANT_VECTOR) {
			zeros_.assign(count ? count : 1, 0);
			f.owned_sel.sel = zeros_.data();
		} else {
			if (dict_.size() < count) throw std::runtime_error("stub: dictionary selection shorter than count");
			f.owned_sel.sel = dict_.data();
		}
		f.sel = &f.owned_sel;
	}
private:
	friend struct FlatVector;
	friend struct ListVector;
	friend struct StructVector;
	LogicalType type_;
	idx_t capacity_;
	VectorType vtype_ = VectorType::FLAT_VECTOR;
	vector<data_t> buffer_;
	ValidityMask validity_;
	vector<unique_ptr<Vector>> child_;
	idx_t list_size_ = 0, list_capacity_ = 0;
	vector<uint32_t> dict_, zeros_;
	vector<unique_ptr<string>> heap_;
	// grow to `cap` entries: a STRUCT grows its fields (a LIST(STRUCT) child), everything else its own buffer
	void Grow(idx_t cap) {
		if (type_.id() == LogicalTypeId::STRUCT) {
			for (auto &c : child_) c->Grow(cap);
		} else {
			buffer_.resize(cap * StubTypeWidth(type_.id()), 0);
		}
		capacity_ = cap;
	}
};
struct FlatVector {
	template <class T>
	static T *GetData(Vector &v) {
		if (v.vtype_ == VectorType::DICTIONARY_VECTOR) throw std::runtime_error("stub: FlatVector::GetData on a dictionary vector");
		return reinterpret_cast<T *>(v.buffer_.data());
	}
	static ValidityMask &Validity(Vector &v) { return v.validity_; }
	static void SetNull(Vector &v, idx_t i, bool is_null) {
		if (i >= v.capacity_) throw std::runtime_error("stub: SetNull beyond the vector's capacity");
		if (is_null) v.validity_.SetInvalid(i); else v.validity_.SetValid(i);
	}
};
struct ListVector {
	static list_entry_t *GetData(Vector &v) { return reinterpret_cast<list_entry_t *>(v.buffer_.data()); }
	static Vector &GetEntry(Vector &v) { return *v.child_[0]; }
	static idx_t GetListSize(const Vector &v) { return v.list_size_; }
	static idx_t GetListCapacity(const Vector &v) { return v.list_capacity_; }
	// writing child entries beyond the reserved capacity is the overrun SURVEY.md §8(b) warns about: the stand-in's child
	// buffer is exactly as large as reserved, so ASan sees it
	static void Reserve(Vector &v, idx_t required) {
		if (required <= v.list_capacity_) return;
		idx_t cap = v.list_capacity_ ? v.list_capacity_ : 16;
		while (cap < required) cap *= 2;
		v.child_[0]->Grow(cap);
		v.list_capacity_ = cap;
	}
	static void SetListSize(Vector &v, idx_t size) {
		if (size > v.list_capacity_) throw std::runtime_error("stub: SetListSize beyond the reserved capacity (ListVector::Reserve missing)");
		v.list_size_ = size;
	}
};
struct StructVector {
	static vector<unique_ptr<Vector>> &GetEntries(Vector &v) { return v.child_; }
};

// ---- functions ----
class ClientContext {};
class Expression {
public:
	explicit Expression(Value v, bool foldable = true) : value_(std::move(v)), foldable_(foldable) {}
	virtual ~Expression() = default;
	virtual bool IsFoldable() const { return foldable_; }
	const Value &StubValue() const { return value_; }
	LogicalType return_type;
private:
	Value value_;
	bool foldable_;
};
struct ExpressionExecutor {
	static Value EvaluateScalar(ClientContext &, const Expression &e) {
		if (!e.IsFoldable()) throw BinderException("stub: expression is not foldable");
		return e.StubValue();
	}
};

struct FunctionData {
	virtual ~FunctionData() = default;
	virtual unique_ptr<FunctionData> Copy() const = 0;
	virtual bool Equals(const FunctionData &other) const = 0;
	template <class T>
	T &Cast() { return dynamic_cast<T &>(*this); } // (DuckDB: reinterpret_cast with a debug check; bad casts must not pass here)
	template <class T>
	const T &Cast() const { return dynamic_cast<const T &>(*this); }
};
class ArenaAllocator {};
enum class AggregateCombineType : uint8_t { PRESERVE_INPUT = 0, ALLOW_DESTRUCTIVE = 1 };
struct AggregateInputData {
	AggregateInputData(optional_ptr<FunctionData> bind_data_p, ArenaAllocator &allocator_p,
	                   AggregateCombineType combine_type_p = AggregateCombineType::PRESERVE_INPUT)
	    : bind_data(bind_data_p), allocator(allocator_p), combine_type(combine_type_p) {}
	optional_ptr<FunctionData> bind_data;
	ArenaAllocator &allocator;
	AggregateCombineType combine_type;
};

class AggregateFunction;
typedef idx_t (*aggregate_size_t)(const AggregateFunction &function);
typedef void (*aggregate_initialize_t)(const AggregateFunction &function, data_ptr_t state);
typedef void (*aggregate_update_t)(Vector inputs[], AggregateInputData &aggr_input_data, idx_t input_count, Vector &state, idx_t count);
typedef void (*aggregate_combine_t)(Vector &state, Vector &combined, AggregateInputData &aggr_input_data, idx_t count);
typedef void (*aggregate_finalize_t)(Vector &state, AggregateInputData &aggr_input_data, Vector &result, idx_t count, idx_t offset);
typedef void (*aggregate_simple_update_t)(Vector inputs[], AggregateInputData &aggr_input_data, idx_t input_count, data_ptr_t state, idx_t count);
typedef unique_ptr<FunctionData> (*bind_aggregate_function_t)(ClientContext &context, AggregateFunction &function,
                                                              vector<unique_ptr<Expression>> &arguments);
typedef void (*aggregate_destructor_t)(Vector &state, AggregateInputData &aggr_input_data, idx_t count);

class AggregateFunction {
public:
	AggregateFunction(const string &name_p, vector<LogicalType> arguments_p, const LogicalType &return_type_p, aggregate_size_t state_size_p,
	                  aggregate_initialize_t initialize_p, aggregate_update_t update_p, aggregate_combine_t combine_p,
	                  aggregate_finalize_t finalize_p, aggregate_simple_update_t simple_update_p = nullptr,
	                  bind_aggregate_function_t bind_p = nullptr, aggregate_destructor_t destructor_p = nullptr)
	    : name(name_p), arguments(std::move(arguments_p)), return_type(return_type_p), state_size(state_size_p), initialize(initialize_p),
	      update(update_p), combine(combine_p), finalize(finalize_p), simple_update(simple_update_p), bind(bind_p), destructor(destructor_p) {}
	template <class STATE>
	static idx_t StateSize(const AggregateFunction &) { return sizeof(STATE); }
	string name;
	vector<LogicalType> arguments;
	LogicalType return_type;
	aggregate_size_t state_size;
	aggregate_initialize_t initialize;
	aggregate_update_t update;
	aggregate_combine_t combine;
	aggregate_finalize_t finalize;
	aggregate_simple_update_t simple_update;
	bind_aggregate_function_t bind;
	aggregate_destructor_t destructor;
};
class AggregateFunctionSet {
public:
	explicit AggregateFunctionSet(const string &name_p) : name(name_p) {}
	void AddFunction(AggregateFunction f) { functions.push_back(std::move(f)); }
	string name;
	vector<AggregateFunction> functions;
};
enum class OnCreateConflict : uint8_t { ERROR_ON_CONFLICT, IGNORE_ON_CONFLICT, REPLACE_ON_CONFLICT, ALTER_ON_CONFLICT };
struct FunctionDescription {
	vector<LogicalType> parameter_types;
	vector<string> parameter_names;
	string description;
	vector<string> examples;
	vector<string> categories;
};
struct CreateAggregateFunctionInfo {
	explicit CreateAggregateFunctionInfo(AggregateFunctionSet set) : functions(std::move(set)) {}
	AggregateFunctionSet functions;
	OnCreateConflict on_conflict = OnCreateConflict::ERROR_ON_CONFLICT;
	string alias_of;
	vector<FunctionDescription> descriptions;
};
class ExtensionLoader {
public:
	void RegisterFunction(CreateAggregateFunctionInfo info) {
		if (registered.count(info.functions.name) && info.on_conflict == OnCreateConflict::ERROR_ON_CONFLICT)
			throw std::runtime_error("stub: function " + info.functions.name + " registered twice");
		const string name = info.functions.name;
		registered.erase(name);
		registered.emplace(name, std::move(info));
	}
	std::map<string, CreateAggregateFunctionInfo> registered;
};

} // namespace duckdb
