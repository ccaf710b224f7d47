"""The C1 input set of SURVEY.md 8(d): the inputs the reference's differential tests feed to its tokenizer.

The reference holds no expected values for them (they come from a JavaScript tokenizer it does not ship), so they
serve as INPUTS: the GPU path must agree with the oracle on every one of them.
  generated()   -- the input families of llama3/internal/testing/vectors.go:18-252 (GenerateTestCases), restated
  comparison()  -- the literal inputs of llama3/comparison_test.go:84-164 (fixture tests/golden/ref_comparison_inputs.json)
"""
import json
import os


def generated():
    c = ["", " ", "\t", "\n", "\r\n", "'", "''", "123456"]                      # edge cases
    for i in range(1, 21):                                                       # runs of blanks around a word
        c += [" " * i + "word", "word" + " " * i]
    c += ["\t" * i + "word" for i in range(1, 11)]
    c += [" \t \t word", "\n\r\n\r\nword", "   \t\t\t   \n\n\n   word", "word   \t\t\t   \n\n\n"]
    for w in ["can", "won", "they", "I", "we", "he", "it"]:                    # contractions in four casings
        for k in ["'s", "'t", "'re", "'ve", "'m", "'ll", "'d"]:
            c += [w + k, w.upper() + k, w + k.upper(), w[:1].upper() + w[1:].lower() + k]
    c += ["0", "00", "000", "0000", "123", "1234", "12345", "1 2 3", "12 34 56", "123 456 789"]
    for p in "!\"#$%&'()*+,-./:;<=>?@[\\]^_`{|}~":
        c += [p, " " + p, p + " "]
    for t in ["café", "naïve", "résumé", "Zürich", "español", "português", "αβγδε", "ΑΒΓΔΕ", "Ελληνικά", "привет", "ПРИВЕТ",
              "Москва", "你好", "世界", "こんにちは", "안녕하세요", "مرحبا", "السلام", "שלום", "עברית", "🦙", "👍", "🌍🌎🌏",
              "👨\u200d👩\u200d👧\u200d👦", "🏳\ufe0f\u200d🌈", "🇺🇸🇬🇧🇯🇵"]:
        c += [t, "Hello " + t + " world"]
    c += [p + "word" for p in "!@#$%^&*()-+=[]{}|\\:;\"<>,.?/"]
    c += ["user@example.com", "first.last@company.co.uk", "https://example.com",
          "http://sub.example.com:8080/path?q=1&v=2#anchor", "192.168.1.1", "2001:0db8:85a3:0000:0000:8a2e:0370:7334",
          "+1-555-123-4567", "$1,234.56", "€999.99", "¥10,000", "50%", "3.14159", "1.23e-4", "#hashtag", "@mention",
          "C:\\Windows\\System32", "/usr/local/bin", "~/Documents/file.txt"]
    c += ["if (x > 0) { return true; }", "func main() { }", "def hello():", "public class Main { }", "const x = () => { };",
          "SELECT * FROM users WHERE id = 1;", "git commit -m 'Initial commit'", "npm install --save-dev @types/node",
          "docker run -p 8080:80 nginx"]
    c += ["a" * 1000, "1" * 1000, " " * 1000, "word " * 200, "'s" * 100]
    return [x.encode("utf-8") for x in c]


def comparison():
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_comparison_inputs.json")
    return [bytes.fromhex(h) for h in json.load(open(path))["inputs"]]
