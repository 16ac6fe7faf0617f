const char afx_build_id_str[] = "eebd15dbe0a7";
