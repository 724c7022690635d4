// Minimal stand-in for the part of <opencv2/core.hpp> the reference's samples use to read their input
// (cv::FileStorage / cv::FileNode on a JSON file, samples/sample_ba_from_file.cpp:91-157).  OpenCV is
// not installed in the build image; with this header on the include path the reference's sample sources
// compile unmodified against libcuda_bundle_adjustment.so.  JSON only, read only.
#pragma once

#include <cstdlib>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#define CV_Assert(expr) do { if (!(expr)) throw std::runtime_error("CV_Assert failed: " #expr); } while (0)

namespace cv
{

namespace detail
{
struct JsonValue
{
	enum Kind { NONE, NUMBER, STRING, ARRAY, OBJECT } kind = NONE;
	double number = 0;
	std::string text;
	std::vector<std::shared_ptr<JsonValue>> items;
	std::map<std::string, std::shared_ptr<JsonValue>> members;
};

class JsonParser
{
public:
	explicit JsonParser(const std::string& s) : s_(s) {}
	std::shared_ptr<JsonValue> parse() { skip(); return value(); }
private:
	void skip() { while (i_ < s_.size() && (s_[i_] == ' ' || s_[i_] == '\n' || s_[i_] == '\r' || s_[i_] == '\t')) i_++; }
	std::shared_ptr<JsonValue> value()
	{
		skip();
		if (i_ >= s_.size()) throw std::runtime_error("json: unexpected end");
		auto v = std::make_shared<JsonValue>();
		const char c = s_[i_];
		if (c == '{')
		{
			v->kind = JsonValue::OBJECT; i_++; skip();
			if (s_[i_] == '}') { i_++; return v; }
			for (;;)
			{
				skip();
				const std::string key = str();
				skip(); expect(':');
				v->members[key] = value();
				skip();
				if (s_[i_] == ',') { i_++; continue; }
				expect('}'); break;
			}
		}
		else if (c == '[')
		{
			v->kind = JsonValue::ARRAY; i_++; skip();
			if (s_[i_] == ']') { i_++; return v; }
			for (;;)
			{
				v->items.push_back(value());
				skip();
				if (s_[i_] == ',') { i_++; continue; }
				expect(']'); break;
			}
		}
		else if (c == '"') { v->kind = JsonValue::STRING; v->text = str(); }
		else if (s_.compare(i_, 4, "true") == 0) { v->kind = JsonValue::NUMBER; v->number = 1; i_ += 4; }
		else if (s_.compare(i_, 5, "false") == 0) { v->kind = JsonValue::NUMBER; v->number = 0; i_ += 5; }
		else if (s_.compare(i_, 4, "null") == 0) { i_ += 4; }
		else
		{
			char* end = nullptr;
			v->kind = JsonValue::NUMBER;
			v->number = std::strtod(s_.c_str() + i_, &end);
			if (end == s_.c_str() + i_) throw std::runtime_error("json: bad number");
			i_ = static_cast<size_t>(end - s_.c_str());
		}
		return v;
	}
	std::string str()
	{
		expect('"');
		std::string out;
		while (i_ < s_.size() && s_[i_] != '"')
		{
			if (s_[i_] == '\\' && i_ + 1 < s_.size()) i_++;
			out.push_back(s_[i_++]);
		}
		expect('"');
		return out;
	}
	void expect(char c)
	{
		if (i_ >= s_.size() || s_[i_] != c) throw std::runtime_error(std::string("json: expected '") + c + "'");
		i_++;
	}
	const std::string& s_;
	size_t i_ = 0;
};
}  // namespace detail

class FileNode
{
public:
	FileNode() = default;
	explicit FileNode(std::shared_ptr<detail::JsonValue> v) : v_(std::move(v)) {}

	FileNode operator[](const char* key) const
	{
		if (!v_ || v_->kind != detail::JsonValue::OBJECT) return FileNode();
		auto it = v_->members.find(key);
		return it == v_->members.end() ? FileNode() : FileNode(it->second);
	}
	FileNode operator[](const std::string& key) const { return (*this)[key.c_str()]; }

	operator int() const { return static_cast<int>(num()); }
	operator float() const { return static_cast<float>(num()); }
	operator double() const { return num(); }
	operator std::string() const { return v_ ? v_->text : std::string(); }
	bool empty() const { return !v_ || v_->kind == detail::JsonValue::NONE; }
	size_t size() const { return v_ ? v_->items.size() : 0; }

	class iterator
	{
	public:
		iterator(const std::vector<std::shared_ptr<detail::JsonValue>>* items, size_t i) : items_(items), i_(i) {}
		FileNode operator*() const { return FileNode((*items_)[i_]); }
		iterator& operator++() { ++i_; return *this; }
		bool operator!=(const iterator& o) const { return i_ != o.i_; }
	private:
		const std::vector<std::shared_ptr<detail::JsonValue>>* items_;
		size_t i_;
	};
	iterator begin() const { return iterator(v_ ? &v_->items : &none(), 0); }
	iterator end() const { return iterator(v_ ? &v_->items : &none(), v_ ? v_->items.size() : 0); }

private:
	double num() const { return v_ && v_->kind == detail::JsonValue::NUMBER ? v_->number : 0.0; }
	static const std::vector<std::shared_ptr<detail::JsonValue>>& none()
	{
		static const std::vector<std::shared_ptr<detail::JsonValue>> e;
		return e;
	}
	std::shared_ptr<detail::JsonValue> v_;
};

class FileStorage
{
public:
	enum Mode { READ = 0 };
	FileStorage() = default;
	FileStorage(const std::string& filename, int /*flags*/) { open(filename); }
	bool open(const std::string& filename)
	{
		std::ifstream in(filename, std::ios::binary);
		if (!in) return false;
		std::stringstream ss;
		ss << in.rdbuf();
		text_ = ss.str();
		size_t start = 0;
		if (text_.compare(0, 5, "%YAML") == 0) start = text_.find('\n');
		body_ = text_.substr(start == std::string::npos ? 0 : start);
		root_ = FileNode(detail::JsonParser(body_).parse());
		opened_ = true;
		return true;
	}
	bool isOpened() const { return opened_; }
	FileNode operator[](const char* key) const { return root_[key]; }
	FileNode operator[](const std::string& key) const { return root_[key]; }
private:
	std::string text_, body_;
	FileNode root_;
	bool opened_ = false;
};

}  // namespace cv
